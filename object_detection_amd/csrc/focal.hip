// CenterNet class loss: label-smoothing focal loss over a softmax across the class axis.
//
// Reference (object_detection/loss.py:8-83, LabelSmoothingFocalLoss.forward / _estimate_difficulty_level):
//   p = softmax(logits, class axis);  difficulty_c = (1 - key_c * p_c)^gamma;  logs = log_softmax(logits)
//   loss = mean over cells of  -sum_c difficulty_c * logs_c * weight_c
//   one_hot_label_format: key = weight = label.long()  (the (N,C,...) target maps, as object_detection/train.py:147-153 uses it)
//   index labels (N,...): key_c = [c == label], weight_c = 1 - smoothing at the label and smoothing/(C-1) elsewhere;
//                         cells whose label equals ignore_index count as zero (loss.py:48-51,66-67)
//   the gradient flows through the difficulty as well (F.softmax(logits) is not detached, loss.py:78-79)
// logits are (N, C, S) with S contiguous.  With one-hot maps almost every cell has no label at all, so the work is split:
//   scan   streams the label maps once (16-byte loads along S) and lists, per workgroup of 1024 cells, the cells that
//          have a label -- each workgroup owns a fixed segment of the list, so no counter is shared and the order is fixed;
//   cells  one wave per listed cell, lanes over the classes: softmax statistics by wave reductions, then the loss terms
//          (forward) or the C gradient values (backward, after the gradient tensor has been zero-filled).
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

struct fl_args {
    const float* logits;     // (N, C, S)
    const void* labels;      // one-hot: (N, C, S) float or int64; index: (N, S) int64
    long cells;              // N * S
    int C;
    long S;
    int one_hot, label_i64;
    float gamma, w_pos, w_neg;
    long ignore_index;
    int has_ignore;
};

__device__ __forceinline__ long label_at(const fl_args& a, size_t i)
{
    return a.label_i64 ? (long)reinterpret_cast<const long long*>(a.labels)[i]
                       : (long)reinterpret_cast<const float*>(a.labels)[i];      // .long(): truncation towards zero
}

// (1 - k p)^gamma and d/d(kp) of it, as torch.pow and its backward give them (gamma == 0: 1 and 0)
__device__ __forceinline__ void difficulty(float base, float gamma, float& d, float& dd)
{
    if (gamma == 0.0f) { d = 1.0f; dd = 0.0f; return; }
    d = powf(base, gamma);
    dd = gamma * powf(base, gamma - 1.0f);
}

// key / weight of class c at one cell
__device__ __forceinline__ void key_weight(const fl_args& a, size_t base, int c, long y, float& k, float& w)
{
    if (a.one_hot) {
        const long t = label_at(a, base + (size_t)c * a.S);
        k = (float)t;
        w = (float)t;
    } else {
        k = c == y ? 1.0f : 0.0f;
        w = c == y ? a.w_pos : a.w_neg;
    }
}

#define FL_REGS 4          // classes per lane held in registers by the per-cell pass (C <= 256)
#define FL_CELLS 1024      // cells per workgroup of the scan = length of its segment of the cell list

// VEC = 4: one-hot fp32 maps, S % 4 == 0, 16-byte aligned: a thread owns four consecutive cells; VEC = 1: anything else
template <int VEC>
__global__ void __launch_bounds__(256) k_focal_scan(const fl_args a, int* __restrict__ list, int* __restrict__ counts)
{
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long first = (long)blockIdx.x * FL_CELLS;
    int filled = 0;
#pragma unroll 1
    for (int round = 0; round < 4 / VEC; ++round) {             // VEC = 4: one round of 256 x 4 cells; VEC = 1: four of 256
        const long cell0 = first + (VEC == 4 ? threadIdx.x * 4l : round * 256l + threadIdx.x);
        bool any[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) any[k] = false;
        if (cell0 < a.cells) {
            if (a.one_hot) {
                const long n = cell0 / a.S, s = cell0 - n * a.S;
                const size_t base = (size_t)n * a.C * a.S + s;
                if (VEC == 4) {
                    const float* __restrict__ lab = reinterpret_cast<const float*>(a.labels) + base;
#pragma unroll 8
                    for (int c = 0; c < a.C; ++c) {
                        const dh_f4 t = __builtin_nontemporal_load(reinterpret_cast<const dh_f4*>(lab + (size_t)c * a.S));
                        any[0] |= (long)t.x != 0;                                       // .long(): |t| >= 1
                        any[VEC > 1 ? 1 : 0] |= (long)t.y != 0;
                        any[VEC > 2 ? 2 : 0] |= (long)t.z != 0;
                        any[VEC > 3 ? 3 : 0] |= (long)t.w != 0;
                    }
                } else {
#pragma unroll 4
                    for (int c = 0; c < a.C; ++c) any[0] |= label_at(a, base + (size_t)c * a.S) != 0;
                }
            } else {
                const long y = label_at(a, (size_t)cell0);
                any[0] = !(a.has_ignore && y == a.ignore_index);                        // loss.py:49-51,66-67
            }
        }
        // ordered compaction of the flagged cells into this workgroup's segment
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const unsigned long long bm = __ballot(any[k]);
            if (lane == 0) s_w[wave] = __popcll(bm);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) before += c; all += c; }
            __syncthreads();
            if (any[k]) list[first + filled + before + __popcll(bm & ((1ull << lane) - 1ull))] = (int)(cell0 + k - first);
            filled += all;
        }
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = filled;
}

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// one workgroup per scan segment, one wave per listed cell.  BWD = false: partial[4*block + wave] receives the wave's
// loss sum; BWD = true: the cell's C gradient values are written (the rest of grad is zero already).
template <bool BWD>
__global__ void __launch_bounds__(256) k_focal_cells(const fl_args a, const int* __restrict__ list, const int* __restrict__ counts,
                                                     double* __restrict__ partial, const float* __restrict__ grad_out,
                                                     float* __restrict__ grad)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long first = (long)blockIdx.x * FL_CELLS;
    const int n_cells = counts[blockIdx.x];
    const float scale = BWD ? grad_out[0] / (float)a.cells : 0.0f;
    double acc = 0.0;
    for (int i = wave; i < n_cells; i += 4) {
        const long cell = first + list[first + i];
        const long n = cell / a.S, s = cell - n * a.S;
        const size_t base = (size_t)n * a.C * a.S + s;
        long y = 0;
        if (!a.one_hot) {
            y = label_at(a, (size_t)cell);
            if (a.has_ignore && y == a.ignore_index) y = 0;
        }
        if (a.C <= 64 * FL_REGS) {
            // the cell's logits and label terms are fetched once (all loads in flight together) and kept in registers
            float xr[FL_REGS], kr[FL_REGS], wr[FL_REGS];
#pragma unroll
            for (int q = 0; q < FL_REGS; ++q) {
                const int c = q * 64 + lane;
                xr[q] = -INFINITY; kr[q] = 0.0f; wr[q] = 0.0f;
                if (c < a.C) {
                    xr[q] = a.logits[base + (size_t)c * a.S];
                    key_weight(a, base, c, y, kr[q], wr[q]);
                }
            }
            float m = xr[0];
#pragma unroll
            for (int q = 1; q < FL_REGS; ++q) m = fmaxf(m, xr[q]);
            m = wave_max_f(m);
            float z = 0.0f;
#pragma unroll
            for (int q = 0; q < FL_REGS; ++q) z += (q * 64 + lane < a.C) ? expf(xr[q] - m) : 0.0f;
            z = dh_wave_sum(z);
            const float lz = logf(z);
            float loss = 0.0f, gsum = 0.0f, g[FL_REGS], pr[FL_REGS];
#pragma unroll
            for (int q = 0; q < FL_REGS; ++q) {
                g[q] = 0.0f; pr[q] = 0.0f;
                if (q * 64 + lane >= a.C) continue;
                const float x = xr[q] - m;
                const float logp = x - lz, p = expf(x) / z;
                pr[q] = p;
                if (wr[q] == 0.0f) continue;
                float d, dd;
                difficulty(1.0f - kr[q] * p, a.gamma, d, dd);
                loss -= d * logp * wr[q];                                               // loss.py:65
                g[q] = wr[q] * (dd * kr[q] * p * logp - d);
                gsum += g[q];
            }
            if (!BWD) {
                acc += (double)dh_wave_sum(loss);
            } else {
                gsum = dh_wave_sum(gsum);
#pragma unroll
                for (int q = 0; q < FL_REGS; ++q)
                    if (q * 64 + lane < a.C) grad[base + (size_t)(q * 64 + lane) * a.S] = scale * (g[q] - pr[q] * gsum);
            }
            continue;
        }
        // more classes than the registers hold: the same arithmetic with the class axis walked once per step
        float m = -INFINITY;
        for (int c = lane; c < a.C; c += 64) m = fmaxf(m, a.logits[base + (size_t)c * a.S]);
        m = wave_max_f(m);
        float z = 0.0f;
        for (int c = lane; c < a.C; c += 64) z += expf(a.logits[base + (size_t)c * a.S] - m);
        z = dh_wave_sum(z);
        const float lz = logf(z);
        float loss = 0.0f, gsum = 0.0f;
        for (int c = lane; c < a.C; c += 64) {
            float k, w;
            key_weight(a, base, c, y, k, w);
            if (w == 0.0f) continue;
            const float x = a.logits[base + (size_t)c * a.S] - m;
            const float logp = x - lz, p = expf(x) / z;
            float d, dd;
            difficulty(1.0f - k * p, a.gamma, d, dd);
            loss -= d * logp * w;                                                       // loss.py:65
            gsum += w * (dd * k * p * logp - d);
        }
        if (!BWD) {
            acc += (double)dh_wave_sum(loss);
        } else {
            gsum = dh_wave_sum(gsum);
            for (int c = lane; c < a.C; c += 64) {
                float k, w;
                key_weight(a, base, c, y, k, w);
                const float x = a.logits[base + (size_t)c * a.S] - m;
                const float logp = x - lz, p = expf(x) / z;
                float g = 0.0f;
                if (w != 0.0f) {
                    float d, dd;
                    difficulty(1.0f - k * p, a.gamma, d, dd);
                    g = w * (dd * k * p * logp - d);
                }
                grad[base + (size_t)c * a.S] = scale * (g - p * gsum);
            }
        }
    }
    if (!BWD && lane == 0) partial[(size_t)blockIdx.x * 4 + wave] = acc;
}

__global__ void __launch_bounds__(1024) k_focal_final(const double* __restrict__ partial, long n_partial, long cells, float* __restrict__ out)
{
    __shared__ double s_p[16];
    double v = 0.0;
    for (long i = threadIdx.x; i < n_partial; i += 1024) v += partial[i];
    v = dh_wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_p[w];
        out[0] = (float)(t / (double)cells);                                           // loss.py:68 (.mean())
    }
}

// zero fill with one non-temporal 16-byte store per thread (n4 vectors)
__global__ void __launch_bounds__(256) k_focal_zero(dh_f4* __restrict__ p, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const dh_f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n4) __builtin_nontemporal_store(z, p + i);
}

struct fl_ws {
    size_t list, counts, partial, total;
    long blocks;
};

fl_ws fl_layout(int64_t cells)
{
    fl_ws w;
    w.blocks = (long)((cells + FL_CELLS - 1) / FL_CELLS);
    size_t at = 0;
    w.list = at;    at += ((size_t)w.blocks * FL_CELLS * sizeof(int) + 255) / 256 * 256;
    w.counts = at;  at += ((size_t)w.blocks * sizeof(int) + 255) / 256 * 256;
    w.partial = at; at += ((size_t)w.blocks * 4 * sizeof(double) + 255) / 256 * 256;
    w.total = at + 256;
    return w;
}

}  // namespace

extern "C" size_t dh_focal_loss_workspace(int64_t cells)
{
    if (cells <= 0) return 0;
    return fl_layout(cells).total;
}

static int focal_args(const float* logits, const void* labels, int64_t N, int C, int64_t S, int one_hot, int label_i64, double gamma,
                      double smoothing, int has_ignore, int64_t ignore_index, fl_args* a)
{
    if (N < 0 || C < 2 || S < 0 || N * S >= (1ll << 40) || gamma < 0) return DH_ERR_ARG;
    if (N * S > 0 && (!logits || !labels)) return DH_ERR_ARG;
    if (!one_hot && !label_i64) return DH_ERR_ARG;
    a->logits = logits; a->labels = labels; a->cells = (long)(N * S); a->C = C; a->S = (long)S;
    a->one_hot = one_hot; a->label_i64 = label_i64; a->gamma = (float)gamma;
    a->w_pos = (float)(1.0 - smoothing);                                               // loss.py:53-56
    a->w_neg = (float)(smoothing / (C - 1));
    a->ignore_index = (long)ignore_index; a->has_ignore = has_ignore;
    return DH_OK;
}

extern "C" int dh_focal_loss_fwd(const float* logits, const void* labels, int64_t N, int C, int64_t S, int one_hot, int label_i64,
                                 double gamma, double smoothing, int has_ignore, int64_t ignore_index, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream)
{
    fl_args a;
    const int rc = focal_args(logits, labels, N, C, S, one_hot, label_i64, gamma, smoothing, has_ignore, ignore_index, &a);
    if (rc != DH_OK) return rc;
    if (!out) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (a.cells == 0) return dh_fill_async(out, 0xFF, sizeof(float), st);   // mean of nothing: NaN
    const fl_ws W = fl_layout(a.cells);
    if (!workspace || workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    if (!dh_aligned16(workspace)) return DH_ERR_ALIGN;
    int* list = (int*)((char*)workspace + W.list);
    int* counts = (int*)((char*)workspace + W.counts);
    double* partial = (double*)((char*)workspace + W.partial);
    const bool vec4 = a.one_hot && !a.label_i64 && a.S % 4 == 0 && dh_aligned16(a.labels);
    if (vec4) hipLaunchKernelGGL(k_focal_scan<4>, dim3((unsigned)W.blocks), dim3(256), 0, st, a, list, counts);
    else hipLaunchKernelGGL(k_focal_scan<1>, dim3((unsigned)W.blocks), dim3(256), 0, st, a, list, counts);
    hipLaunchKernelGGL(k_focal_cells<false>, dim3((unsigned)W.blocks), dim3(256), 0, st, a, (const int*)list, (const int*)counts, partial,
                       (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_focal_final, dim3(1), dim3(1024), 0, st, (const double*)partial, W.blocks * 4, a.cells, out);
    return dh_launch_status();
}

extern "C" int dh_focal_loss_bwd(const float* logits, const void* labels, int64_t N, int C, int64_t S, int one_hot, int label_i64,
                                 double gamma, double smoothing, int has_ignore, int64_t ignore_index, const float* grad_out,
                                 float* grad_logits, const void* workspace, void* stream)
{
    fl_args a;
    const int rc = focal_args(logits, labels, N, C, S, one_hot, label_i64, gamma, smoothing, has_ignore, ignore_index, &a);
    if (rc != DH_OK) return rc;
    if (a.cells == 0) return DH_OK;
    if (!grad_out || !grad_logits || !workspace) return DH_ERR_ARG;
    const fl_ws W = fl_layout(a.cells);
    hipStream_t st = (hipStream_t)stream;
    const size_t n_el = (size_t)a.cells * C;
    if (n_el % 4 == 0 && dh_aligned16(grad_logits) && n_el / 4 / 256 < (1ull << 31)) {
        hipLaunchKernelGGL(k_focal_zero, dim3((unsigned)((n_el / 4 + 255) / 256)), dim3(256), 0, st, (dh_f4*)grad_logits, n_el / 4);
    } else if (dh_fill_async(grad_logits, 0, sizeof(float) * n_el, st) != DH_OK) {
        return DH_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k_focal_cells<true>, dim3((unsigned)W.blocks), dim3(256), 0, st, a,
                       (const int*)((const char*)workspace + W.list), (const int*)((const char*)workspace + W.counts),
                       (double*)nullptr, grad_out, grad_logits);
    return dh_launch_status();
}
