// YOLOv5 ComputeLoss forward and backward over all levels.
//
// Reference: object_detection/yolov5/utils/loss.py:117-167 (ComputeLoss.__call__), :35-62 (FocalLoss),
//            torch.nn.BCEWithLogitsLoss(pos_weight) for the element formula.
//   per level i:  ps = pi[b,a,gj,gi];  pxy = 2s-0.5, pwh = (2s)^2*anchor;  iou = CIoU(pbox, tbox)
//                 lbox += mean(1-iou);  tobj[b,a,gj,gi] = (1-gr) + gr*clamp(iou,0)  (duplicates: last row wins)
//                 lcls += mean BCE(ps[:,5:], onehot(cp,cn));  lobj += mean BCE(pi[...,4], tobj) * balance[i]
//   loss = (lbox*box + lobj*obj + lcls*cls) * bs
//
// Forward = 3 launches: positives (one wave per assigned row), dense objectness (reads the strided
// obj logit of every anchor + a 4-byte word of the cell map), finalisation.  The dense tobj tensor is never
// built: a per-cell map stands in for it.  Each assigned row exchanges its id into its cell's word and keeps
// what it found there, so the rows of a cell form a list (usually of one); the cell's target comes from the
// largest row id in it -- the reference's "last duplicate wins" (loss.py:143) -- whatever order the exchanges
// happened in.  The objectness pass also stores d BCE / d logit per anchor (4 B each), so the backward never
// re-reads the strided logits.  Backward = 2 launches: one streaming pass writes the whole dense gradient
// (zeros + scaled objectness term, 16-byte stores), then, per cell with assigned rows, the cell's last row sums
// the box / class terms of all of them in row order and stores the sums: the gradient is the same bit for bit
// on every run, and the order is that of a sequential index_put(accumulate=True).
//
// Convolution layout (the *_conv entries): the maps are the outputs of the head's 1x1 convolutions as they are,
// (bs, na*no, ny, nx), i.e. what Detect.forward holds before the permuted copy of object_detection/yolov5/models/yolo.py:54,
// and the gradient is produced in that layout too, so neither that copy nor its backward has to exist.  Cells keep their
// numbering ((b*na + a)*ny + gj)*nx + gi; channel c of a cell sits ny*nx elements after channel c - 1.  The objectness
// logits are then contiguous planes (coalesced reads without a compact copy) and the dense gradient is zero planes plus
// one scaled plane per (image, anchor).
#include "dh_iou.h"
#include "dh_internal.h"
#include "../../include/dethead.h"

namespace {

struct yl_level {
    const float* p;      // (bs, na, ny, nx, no)
    float* g;            // gradient, same shape (backward only)
    int ny, nx;
    long cells;          // bs*na*ny*nx
    long cell_first;     // offset of this level in the per-cell map (one word per cell: the head of the cell's row list, -1 if none)
    dh_fastdiv per_image;   // na*ny*nx
    dh_fastdiv plane;       // ny*nx
    dh_fastdiv gchunks;     // convolution-layout dense gradient: workgroups per (image, anchor) span
    uint32_t row_first;  // first row of this level inside one image of the decoded (bs, A) planes
    unsigned blocks;     // dense-pass blocks of this level (forward: objectness blocks, backward: gradient blocks)
    unsigned fwd_blocks; // objectness-pass blocks of this level: one partial sum each
    float balance;
};

struct yl_params {
    yl_level lvl[DH_MAX_LEVELS];
    int nl, bs, na, nc, no;
    int half;                  // convolution layout only: the tensors hold fp16 elements
    int half_grad;             // ... and the gradient is written in fp16 too (else fp32)
    int conv;                  // 0: maps (bs, na, ny, nx, no); 1: convolution layout (bs, na*no, ny, nx)
    uint32_t rows_per_image;   // A = sum over levels of na*ny*nx
    long cap;
    float box, obj, cls, cls_pw, obj_pw, gamma, cp, cn, gr;
    int sort_iou;              // the best IoU among a cell's rows owns the cell instead of its last row (loss.py:140-142)
    const float* level_scale;  // backward only, may be null: per level [factor on the box / class terms, factor on the objectness term]
};

#define FOCAL_ALPHA 0.25f   // default alpha of FocalLoss (loss.py:37); ComputeLoss never overrides it

// BCEWithLogits(x, t, pos_weight pw), optionally wrapped by FocalLoss(gamma)
__device__ __forceinline__ float bce_value(float x, float t, float pw, float gamma)
{
    const float lw = (pw - 1.0f) * t + 1.0f;
    float l = (1.0f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
    if (gamma > 0.0f) {
        const float p = 1.0f / (1.0f + expf(-x));
        const float pt = t * p + (1.0f - t) * (1.0f - p);
        const float af = t * FOCAL_ALPHA + (1.0f - t) * (1.0f - FOCAL_ALPHA);
        l *= af * powf(1.0f - pt, gamma);
    }
    return l;
}

__device__ __forceinline__ float bce_grad(float x, float t, float pw, float gamma)
{
    const float lw = (pw - 1.0f) * t + 1.0f;
    const float s = 1.0f / (1.0f + expf(-x));
    const float d = lw * s - pw * t;          // = (1-t) - lw*(1-s) without the cancellation at small s
    if (!(gamma > 0.0f)) return d;
    // FocalLoss (loss.py:45-62) differentiated.  Evaluated in the order of the forward formula, the loss value
    // x + (softplus(-|x|) - x) and the modulating base 1 - (1 - s) lose 1e-4 of their value to rounding at the usual
    // negative logits (the fp32 reference carries that noise too).  Both are formed here from terms of one sign:
    //   l      = lw*softplus(-|x|) + (x < 0 ? |x|*pw*t : (1-t)*x)
    //   1 - pt = t*(1-s) + (1-t)*s,   1 - s = sigmoid(-x)
    const float e = expf(-fabsf(x));
    const float sp = log1pf(e);
    const float l = x < 0.0f ? lw * sp + (-x) * (pw * t) : (1.0f - t) * x + lw * sp;
    const float sn = x < 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
    const float af = t * FOCAL_ALPHA + (1.0f - t) * (1.0f - FOCAL_ALPHA);
    const float om = t * sn + (1.0f - t) * s;
    const float mf = powf(om, gamma);
    const float dmf = -gamma * powf(om, gamma - 1.0f) * ((2.0f * t - 1.0f) * s * sn);
    return af * (d * mf + l * dmf);
}

// element k of a tensor whose elements are fp32 or fp16 (widened on load; the arithmetic is fp32 either way)
template <typename T> __device__ __forceinline__ float ld_el(const float* base, size_t k)
{
    return (float)reinterpret_cast<const T*>(base)[k];
}
template <typename T> __device__ __forceinline__ float ld_el_nt(const float* base, size_t k)
{
    return (float)__builtin_nontemporal_load(reinterpret_cast<const T*>(base) + k);
}

// gradient element k of a tensor of fp32 or fp16 elements += v.  fp16: the hardware's packed add on the aligned pair that
// holds the element (the other half adds zero); every add rounds to fp16
typedef _Float16 dh_h2v __attribute__((ext_vector_type(2)));
typedef _Float16 dh_h4v __attribute__((ext_vector_type(4)));
template <typename G> __device__ __forceinline__ void add_el(float* base, size_t k, float v);
template <> __device__ __forceinline__ void add_el<float>(float* base, size_t k, float v) { atomicAdd(base + k, v); }
template <> __device__ __forceinline__ void add_el<_Float16>(float* base, size_t k, float v)
{
    _Float16* h = reinterpret_cast<_Float16*>(base);
    const _Float16 hv = (_Float16)v, zero = (_Float16)0.0f;
    const dh_h2v pair = (k & 1) ? dh_h2v{zero, hv} : dh_h2v{hv, zero};
    __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<dh_h2v*>(h + (k & ~(size_t)1)), pair);
}

struct pos_row {
    long cell;     // index of (b,a,gj,gi) inside the level
    size_t off;    // element offset of channel 0 of that cell inside the level's tensor
    uint32_t cs;   // distance between consecutive channels of the cell
    int cls;
    float tb[4], an[2];
    bool ok;
};

template <bool CONV>
__device__ __forceinline__ pos_row load_pos(const yl_params& P, const yl_level& L, int l, long r,
                                            const long long* __restrict__ idx, const float* __restrict__ tbox,
                                            const float* __restrict__ anch, const long long* __restrict__ tcls)
{
    pos_row R;
    const long long* base = idx + (size_t)l * 4 * P.cap;
    const long long b = base[r], a = base[P.cap + r], gj = base[2 * P.cap + r], gi = base[3 * P.cap + r];
    R.ok = b >= 0 && b < P.bs && a >= 0 && a < P.na && gj >= 0 && gj < L.ny && gi >= 0 && gi < L.nx;
    R.cell = ((b * P.na + a) * L.ny + gj) * L.nx + gi;
    if (CONV) {
        const uint32_t hw = L.plane.d;
        R.off = (size_t)(b * P.na + a) * P.no * hw + (size_t)(gj * L.nx + gi);
        R.cs = hw;
    } else {
        R.off = (size_t)R.cell * P.no;
        R.cs = 1;
    }
    const float4 tb = *reinterpret_cast<const float4*>(tbox + ((size_t)l * P.cap + r) * 4);
    R.tb[0] = tb.x; R.tb[1] = tb.y; R.tb[2] = tb.z; R.tb[3] = tb.w;
    const float2 an = *reinterpret_cast<const float2*>(anch + ((size_t)l * P.cap + r) * 2);
    R.an[0] = an.x; R.an[1] = an.y;
    R.cls = (int)tcls[(size_t)l * P.cap + r];
    return R;
}

// ------------------------------------------------------------------------------------ forward: positives
template <bool CONV, typename T = float>
__global__ void __launch_bounds__(256) k_yl_pos_fwd(const yl_params P, const long long* __restrict__ idx,
                                                    const float* __restrict__ tbox, const float* __restrict__ anch,
                                                    const long long* __restrict__ tcls, const int* __restrict__ counts,
                                                    int* __restrict__ owner, int* __restrict__ nxt, float* __restrict__ iou_ws,
                                                    float* __restrict__ cls_ws, int* __restrict__ invalid)
{
    const int l = blockIdx.y;
    const yl_level& L = P.lvl[l];
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= counts[l]) return;
    const pos_row R = load_pos<CONV>(P, L, l, r, idx, tbox, anch, tcls);
    float* __restrict__ iou_out = iou_ws + (size_t)l * P.cap;
    float* __restrict__ cls_out = cls_ws + (size_t)l * P.cap;
    if (!R.ok) {      // image / anchor index outside the tensors: contributes nothing, counted for the caller
        if (lane == 0) { iou_out[r] = 1.0f; cls_out[r] = 0.0f; atomicAdd(invalid, 1); }
        return;
    }
    const size_t cs = CONV ? R.cs : 1;      // a compile-time 1 for the map layout: channel offsets stay immediates
    const float* __restrict__ ps = L.p + (sizeof(T) == 4 ? R.off : 0);       // fp16: element offsets are applied by ld_el
    const size_t e0 = sizeof(T) == 4 ? 0 : R.off;
    const float x0 = ld_el<T>(ps, e0), x1 = ld_el<T>(ps, e0 + cs), x2 = ld_el<T>(ps, e0 + 2 * cs), x3 = ld_el<T>(ps, e0 + 3 * cs);
    const float px = dh_sigmoid(x0) * 2.0f - 0.5f, py = dh_sigmoid(x1) * 2.0f - 0.5f;          // loss.py:132
    const float tw = dh_sigmoid(x2) * 2.0f, th = dh_sigmoid(x3) * 2.0f;
    const float pw = (tw * tw) * R.an[0], ph = (th * th) * R.an[1];                             // loss.py:133
    const float iou = dh_iou_value<DH_CIOU>(dh_from_xywh(px, py, pw, ph), dh_from_xywh(R.tb[0], R.tb[1], R.tb[2], R.tb[3]),
                                            1e-7f, (float)(1.0 + 1e-7));
    float csum = 0.0f;
    if (P.nc > 1) {                                                                             // loss.py:146-149
        for (int c = lane; c < P.nc; c += 64)
            csum += bce_value(ld_el<T>(ps, e0 + (5 + c) * cs), c == R.cls ? P.cp : P.cn, P.cls_pw, P.gamma);
        csum = dh_wave_sum(csum);
    }
    if (lane == 0) {
        iou_out[r] = iou;
        cls_out[r] = csum;
        // the row joins its cell's list: the cell's word names the row that arrived last, every row the one before it
        nxt[(size_t)l * P.cap + r] = atomicExch(owner + L.cell_first + R.cell, (int)r);
    }
}

// ------------------------------------------------------------------------------------ forward: dense objectness
__device__ __forceinline__ int level_of_block(const yl_params& P, unsigned& blk)
{
    int l = 0;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k)
        if (l == k && k + 1 < P.nl && blk >= P.lvl[k].blocks) { blk -= P.lvl[k].blocks; l = k + 1; }
    return l;
}

// The largest row id in the list that starts at `head` (-1: none): the row whose value index_put_ leaves in the cell
// (loss.py:143).  The walk is bounded by the row capacity, whatever the words hold.
__device__ __forceinline__ int last_row_of(int head, const int* __restrict__ nxt_l, long cap)
{
    int m = head;
    long steps = 0;
    for (int j = head; j >= 0 && steps < cap; ++steps) {
        m = j > m ? j : m;
        j = nxt_l[j];
    }
    return m;
}

__device__ __forceinline__ float tobj_of(const yl_params& P, int head, const float* __restrict__ iou_l,
                                         const int* __restrict__ nxt_l)
{
    if (head < 0) return 0.0f;
    float score;
    if (P.sort_iou) {            // rows sorted by ascending clamped IoU before index_put_: the largest one is written last
        score = 0.0f;            // (loss.py:139-143); equal scores write equal values, so their order does not matter
        long steps = 0;
        for (int j = head; j >= 0 && steps < P.cap; ++steps) {
            score = fmaxf(score, iou_l[j]);
            j = nxt_l[j];
        }
    } else {
        score = fmaxf(iou_l[last_row_of(head, nxt_l, P.cap)], 0.0f);                              // loss.py:139
    }
    return (1.0f - P.gr) + P.gr * score;                                                          // loss.py:143
}

#define OBJ_CELLS_PER_BLOCK 1024

// VEC: a thread owns four consecutive cells (same image: the cells per image and level are a multiple of four) and
// moves the compact logits, the cells' words and the stored derivative as 16-byte vectors.
template <bool VEC, bool CONV = false, typename T = float>
__global__ void __launch_bounds__(256) k_yl_obj_fwd(const yl_params P, const int* __restrict__ owner,
                                                    const int* __restrict__ nxt, const float* __restrict__ iou_ws,
                                                    double* __restrict__ partial, float* __restrict__ dobj,
                                                    const float* __restrict__ logits)
{
    unsigned blk = blockIdx.x;
    const int l = level_of_block(P, blk);
    const yl_level& L = P.lvl[l];
    const float* __restrict__ iou_l = iou_ws + (size_t)l * P.cap;
    const int* __restrict__ nxt_l = nxt + (size_t)l * P.cap;
    const int* __restrict__ own_l = owner + L.cell_first;
    double acc = 0.0;
    if (VEC) {
        const long cell = (long)blk * OBJ_CELLS_PER_BLOCK + threadIdx.x * 4;
        if (cell < L.cells) {
            const uint32_t b = dh_div((uint32_t)cell, L.per_image);
            const dh_f4 x4 = *reinterpret_cast<const dh_f4*>(
                logits + (size_t)b * P.rows_per_image + L.row_first + ((uint32_t)cell - b * L.per_image.d));
            const int4 o4 = *reinterpret_cast<const int4*>(own_l + cell);
            const float x[4] = {x4.x, x4.y, x4.z, x4.w};
            const int o[4] = {o4.x, o4.y, o4.z, o4.w};
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = tobj_of(P, o[k], iou_l, nxt_l);
                acc += (double)bce_value(x[k], t, P.obj_pw, P.gamma);                              // loss.py:155
                d[k] = bce_grad(x[k], t, P.obj_pw, P.gamma);
            }
            const dh_f4 d4 = {d[0], d[1], d[2], d[3]};
            *reinterpret_cast<dh_f4*>(dobj + L.cell_first + cell) = d4;
        }
    } else {
#pragma unroll
        for (int u = 0; u < OBJ_CELLS_PER_BLOCK / 256; ++u) {
            const long cell = (long)blk * OBJ_CELLS_PER_BLOCK + u * 256 + threadIdx.x;
            if (cell < L.cells) {
                float x;
                if (logits) {      // compact copy of the logits in decoded-row order: (bs, A)
                    const uint32_t b = dh_div((uint32_t)cell, L.per_image);
                    x = logits[(size_t)b * P.rows_per_image + L.row_first + ((uint32_t)cell - b * L.per_image.d)];
                } else if (CONV) {     // the objectness plane of the cell's (image, anchor): consecutive cells, consecutive words
                    const uint32_t ba = dh_div((uint32_t)cell, L.plane);
                    x = ld_el_nt<T>(L.p, ((size_t)ba * P.no + 4) * L.plane.d + ((uint32_t)cell - ba * L.plane.d));
                } else {
                    // one 4-byte word per 340-byte row, touched once: a non-temporal load fetches just its 64-byte sector
                    // and keeps it out of the caches (measured 2.2x faster than a default load on a cold tensor)
                    x = __builtin_nontemporal_load(L.p + (size_t)cell * P.no + 4);
                }
                const float t = tobj_of(P, own_l[cell], iou_l, nxt_l);
                acc += (double)bce_value(x, t, P.obj_pw, P.gamma);                                 // loss.py:155
                // the backward needs d BCE / d logit of every anchor: keep it compact (4 B per anchor) so that
                // the gradient pass does not have to touch the 340-byte-strided logits again
                dobj[L.cell_first + cell] = bce_grad(x, t, P.obj_pw, P.gamma);
            }
        }
    }
    __shared__ double red[4];
    acc = dh_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out: [0] loss, [1..3] lbox,lobj,lcls (after gains), [4] assigned rows over all levels,
//      [5..5+nl) mean objectness BCE per level, [5+nl] ignored rows
#define FINAL_THREADS 1024

// the reduction proper, run by THREADS threads of one workgroup
template <int THREADS>
__device__ __forceinline__ void yl_finalize(const yl_params& P, const int* __restrict__ counts, const float* __restrict__ iou_ws,
                                            const float* __restrict__ cls_ws, const double* __restrict__ partial,
                                            const int* __restrict__ invalid, float* __restrict__ out,
                                            double* __restrict__ sums, double (*red)[THREADS / 64])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned first_block = 0;
    for (int l = 0; l < P.nl; ++l) {       // all loads of all levels are independent: no barrier inside the loop
        const int n = counts[l];
        double sb = 0.0, sc = 0.0, so = 0.0;
        for (int r = threadIdx.x; r < n; r += THREADS) {
            sb += (double)(1.0f - iou_ws[(size_t)l * P.cap + r]);
            sc += (double)cls_ws[(size_t)l * P.cap + r];
        }
        for (unsigned k = threadIdx.x; k < P.lvl[l].fwd_blocks; k += THREADS) so += partial[first_block + k];
        first_block += P.lvl[l].fwd_blocks;
        sb = dh_wave_sum_d(sb); sc = dh_wave_sum_d(sc); so = dh_wave_sum_d(so);
        if (lane == 0) { red[3 * l + 0][wave] = sb; red[3 * l + 1][wave] = sc; red[3 * l + 2][wave] = so; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
        long n_pos = 0;
        for (int l = 0; l < P.nl; ++l) {
            n_pos += counts[l];
            double tb = 0.0, tc = 0.0, to = 0.0;
            for (int w = 0; w < THREADS / 64; ++w) { tb += red[3 * l][w]; tc += red[3 * l + 1][w]; to += red[3 * l + 2][w]; }
            const int n = counts[l];
            const float s_box = n > 0 ? (float)(tb / n) : 0.0f;
            const float s_cls = (n > 0 && P.nc > 1) ? (float)(tc / ((double)n * P.nc)) : 0.0f;
            const float s_obj = (float)(to / (double)P.lvl[l].cells);
            lbox += s_box;
            lcls += s_cls;
            lobj += s_obj * P.lvl[l].balance;                                                    // loss.py:156
            out[5 + l] = s_obj;
            // the un-normalised sums and their divisors: what ranks add up to form the means of the whole batch
            sums[5 * l + 0] = tb; sums[5 * l + 1] = P.nc > 1 ? tc : 0.0; sums[5 * l + 2] = to;
            sums[5 * l + 3] = (double)n; sums[5 * l + 4] = (double)P.lvl[l].cells;
        }
        lbox *= P.box; lobj *= P.obj; lcls *= P.cls;                                             // loss.py:162-164
        out[0] = (lbox + lobj + lcls) * (float)P.bs;
        out[1] = lbox; out[2] = lobj; out[3] = lcls;
        out[4] = (float)n_pos;
        out[5 + P.nl] = (float)(invalid[0] + 1);
    }
}

__global__ void __launch_bounds__(FINAL_THREADS) k_yl_final(const yl_params P, const int* __restrict__ counts,
                                                            const float* __restrict__ iou_ws, const float* __restrict__ cls_ws,
                                                            const double* __restrict__ partial, const int* __restrict__ invalid,
                                                            float* __restrict__ out, double* __restrict__ sums)
{
    __shared__ double red[3 * DH_MAX_LEVELS][FINAL_THREADS / 64];
    yl_finalize<FINAL_THREADS>(P, counts, iou_ws, cls_ws, partial, invalid, out, sums, red);
}

// ------------------------------------------------------------------------------------ backward: dense pass
// Writes every element of the gradient once: zero everywhere except the objectness channel.
#define GRAD_VEC_PER_BLOCK 256

// FINAL: workgroup 0 also reduces the forward's partial sums into `out` (the fused forward+backward entry: the
// reduction then costs nothing, hidden under this bandwidth-bound launch, instead of a launch of its own).
struct yl_final_args {
    const int* counts;
    const float* iou_ws;
    const float* cls_ws;
    const double* partial;
    const int* invalid;
    float* out;
    double* sums;
};

template <int VEC, bool FINAL>
__global__ void __launch_bounds__(256) k_yl_obj_bwd(const yl_params P, const float* __restrict__ dobj,
                                                    const float* __restrict__ g_loss, const dh_fastdiv nod,
                                                    const yl_final_args F)
{
    if (FINAL && blockIdx.x == 0) {
        __shared__ double red[3 * DH_MAX_LEVELS][4];
        yl_finalize<256>(P, F.counts, F.iou_ws, F.cls_ws, F.partial, F.invalid, F.out, F.sums, red);
    }
    unsigned blk = blockIdx.x;
    const int l = level_of_block(P, blk);
    const yl_level& L = P.lvl[l];
    const float* __restrict__ d_l = dobj + L.cell_first;
    const float scale = g_loss[0] * (float)P.bs * P.obj * L.balance / (float)L.cells * (P.level_scale ? P.level_scale[2 * l + 1] : 1.0f);
    const size_t total = (size_t)L.cells * P.no;
    const size_t nvec = total / VEC;
#pragma unroll
    for (int u = 0; u < GRAD_VEC_PER_BLOCK / 256; ++u) {
        const size_t v = (size_t)blk * GRAD_VEC_PER_BLOCK + u * 256 + threadIdx.x;
        if (v >= nvec) continue;
        const size_t e = v * VEC;
        // position of the objectness element (channel 4) relative to this vector
        const uint32_t cell = dh_div((uint32_t)e, nod);          // total < 2^32 is checked on the host
        const uint32_t c0 = (uint32_t)e - cell * nod.d;
        float o[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = 0.0f;
        // channel 4 of `cell` sits at offset 4 - c0; of the next cell at 4 - c0 + no
        int k4 = 4 - (int)c0;
        uint32_t hit = cell;
        if (k4 < 0) { k4 += (int)nod.d; hit = cell + 1; }
        if (k4 < VEC && hit < (uint32_t)L.cells) {
            const float g = scale * d_l[hit];
#pragma unroll
            for (int k = 0; k < VEC; ++k) if (k == k4) o[k] = g;
        }
        if (VEC == 4) {      // write-once stream: keep it out of the caches the next kernels want
            const dh_f4 ov = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(ov, reinterpret_cast<dh_f4*>(L.g + e));
        } else {
            L.g[e] = o[0];
        }
    }
}

// The dense pass for the convolution layout: per (image, anchor) a span of no planes of ny*nx elements, all zero except
// plane 4, which is the stored objectness derivative times the level's scale.  A workgroup owns 256 vectors of one span,
// so whether a vector lies in the objectness plane is a range test (ny*nx % VEC == 0: no vector straddles planes).
template <int VEC, bool FINAL, typename G = float>
__global__ void __launch_bounds__(256) k_yl_obj_bwd_conv(const yl_params P, const float* __restrict__ dobj,
                                                         const float* __restrict__ g_loss, const yl_final_args F)
{
    if (FINAL && blockIdx.x == 0) {
        __shared__ double red[3 * DH_MAX_LEVELS][4];
        yl_finalize<256>(P, F.counts, F.iou_ws, F.cls_ws, F.partial, F.invalid, F.out, F.sums, red);
    }
    unsigned blk = blockIdx.x;
    const int l = level_of_block(P, blk);
    const yl_level& L = P.lvl[l];
    const uint32_t hw = L.plane.d;
    const uint32_t span = (uint32_t)P.no * hw;
    const uint32_t ba = dh_div(blk, L.gchunks);
    const uint32_t e = ((blk - ba * L.gchunks.d) * 256 + threadIdx.x) * VEC;
    if (e >= span) return;
    const float scale = g_loss[0] * (float)P.bs * P.obj * L.balance / (float)L.cells * (P.level_scale ? P.level_scale[2 * l + 1] : 1.0f);
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = 0.0f;
    if (e >= 4 * hw && e < 5 * hw) {
        const float* __restrict__ d = dobj + L.cell_first + (size_t)ba * hw + (e - 4 * hw);
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = scale * d[k];
    }
    G* __restrict__ g = reinterpret_cast<G*>(L.g) + (size_t)ba * span + e;
    if (VEC == 4) {
        if (sizeof(G) == 4) {
            const dh_f4 ov = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(ov, reinterpret_cast<dh_f4*>(g));
        } else {
            const dh_h4v ov = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            __builtin_nontemporal_store(ov, reinterpret_cast<dh_h4v*>(g));
        }
    } else {
        g[0] = (G)o[0];
    }
}

// ------------------------------------------------------------------------------------ backward: positives
// One wave per assigned row.  The rows of a cell (its list, see k_yl_pos_fwd) are handled by the cell's last row alone:
// lane i of that wave takes the list's i-th row, the rows are ranked by id, and their box / class terms are summed in
// ascending row order -- what a sequential index_put(accumulate=True) over the rows would add up -- and stored (the dense
// pass has zero-filled these elements; nobody else writes them).  The other rows of the cell have nothing to do.
// A cell shared by more than POS_MATES rows (hundreds of targets on one grid cell and anchor) falls back to one
// floating-point atomic per row and element, whose order is not fixed.
#define POS_MATES 64

template <typename G> __device__ __forceinline__ void put_el(float* base, size_t k, float v, bool atomic);
template <> __device__ __forceinline__ void put_el<float>(float* base, size_t k, float v, bool atomic)
{
    if (atomic) atomicAdd(base + k, v); else base[k] = v;
}
template <> __device__ __forceinline__ void put_el<_Float16>(float* base, size_t k, float v, bool atomic)
{
    if (atomic) add_el<_Float16>(base, k, v); else reinterpret_cast<_Float16*>(base)[k] = (_Float16)v;
}

// d loss / d logit of box channel `lane` (0..3) from one row: target box b, anchor (a0, a1); s = sigmoid of the four logits
__device__ __forceinline__ float box_term(int lane, float s0, float s1, float s2, float s3, float a0, float a1, float b0, float b1,
                                          float b2, float b3, float g_iou)
{
    const float tw = s2 * 2.0f, th = s3 * 2.0f;
    const dh_corners a = dh_from_xywh(s0 * 2.0f - 0.5f, s1 * 2.0f - 0.5f, (tw * tw) * a0, (th * th) * a1);
    dh_corners gc;
    dh_iou_value_grad<DH_CIOU>(a, dh_from_xywh(b0, b1, b2, b3), 1e-7f, (float)(1.0 + 1e-7), g_iou, gc);
    float gx, gy, gw, gh;
    dh_grad_to_xywh(gc, gx, gy, gw, gh);
    if (lane == 0) return gx * 2.0f * s0 * (1.0f - s0);
    if (lane == 1) return gy * 2.0f * s1 * (1.0f - s1);
    if (lane == 2) return gw * (2.0f * tw * a0) * (2.0f * s2 * (1.0f - s2));
    return gh * (2.0f * th * a1) * (2.0f * s3 * (1.0f - s3));
}

template <bool CONV, typename T = float, typename G = float>
__global__ void __launch_bounds__(256) k_yl_pos_bwd(const yl_params P, const long long* __restrict__ idx,
                                                    const float* __restrict__ tbox, const float* __restrict__ anch,
                                                    const long long* __restrict__ tcls, const int* __restrict__ counts,
                                                    const float* __restrict__ g_loss, const int* __restrict__ owner,
                                                    const int* __restrict__ nxt)
{
    const int l = blockIdx.y;
    const yl_level& L = P.lvl[l];
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = counts[l];
    if (r >= n) return;
    const pos_row R = load_pos<CONV>(P, L, l, r, idx, tbox, anch, tcls);
    if (!R.ok) return;
    const float* __restrict__ ps = L.p + (sizeof(T) == 4 ? R.off : 0);
    const size_t e0 = sizeof(T) == 4 ? 0 : R.off;
    float* __restrict__ gs = L.g + (sizeof(G) == 4 ? R.off : 0);      // fp16 gradient: put_el applies element offsets
    const size_t g0 = sizeof(G) == 4 ? 0 : R.off;
    const size_t cs = CONV ? R.cs : 1;
    // The cell's logits do not depend on the list, and neither does this row's own link: all these loads go out together,
    // so the usual case -- the row is alone on its cell -- costs no round trip more than reading the row itself.
    const float x0 = ld_el<T>(ps, e0), x1 = ld_el<T>(ps, e0 + cs), x2 = ld_el<T>(ps, e0 + 2 * cs), x3 = ld_el<T>(ps, e0 + 3 * cs);
    const float xc0 = lane < P.nc ? ld_el<T>(ps, e0 + (5 + lane) * cs) : 0.0f;
    const float xc1 = lane + 64 < P.nc ? ld_el<T>(ps, e0 + (5 + lane + 64) * cs) : 0.0f;
    const int* __restrict__ nxt_l = nxt + (size_t)l * P.cap;
    const int head = owner[L.cell_first + R.cell];
    const int before_me = nxt_l[r];

    const float up = g_loss[0] * (float)P.bs * (P.level_scale ? P.level_scale[2 * l] : 1.0f);
    const float s0 = dh_sigmoid(x0), s1 = dh_sigmoid(x1), s2 = dh_sigmoid(x2), s3 = dh_sigmoid(x3);
    const float g_iou = -up * P.box / (float)n;                  // d lbox / d iou = -box/n  (lbox += mean(1 - iou))
    const float sc = up * P.cls / ((float)n * (float)P.nc);
    if (head == (int)r && before_me < 0) {
        // alone on its cell: the row's own terms, stored
        if (lane < 4)
            put_el<G>(gs, g0 + lane * cs, box_term(lane, s0, s1, s2, s3, R.an[0], R.an[1], R.tb[0], R.tb[1], R.tb[2], R.tb[3], g_iou), false);
        if (P.nc > 1) {
            for (int c0 = 0; c0 < P.nc; c0 += 64) {
                const int c = c0 + lane;
                const float x = c0 == 0 ? xc0 : (c0 == 64 ? xc1 : (c < P.nc ? ld_el<T>(ps, e0 + (5 + c) * cs) : 0.0f));
                if (c < P.nc) put_el<G>(gs, g0 + (5 + c) * cs, sc * bce_grad(x, c == R.cls ? P.cp : P.cn, P.cls_pw, P.gamma), false);
            }
        }
        return;
    }
    // the cell's rows: lane i keeps the i-th of the list; `last` is the largest id, `k` their number
    int k = 0, last = -1, mine = -1;
    for (int j = head; j >= 0 && k <= (int)P.cap; j = nxt_l[j]) {
        if (lane == k) mine = j;
        last = j > last ? j : last;
        ++k;
    }
    const bool shared = k <= POS_MATES;
    if (shared && last != (int)r) return;                     // wave-uniform: this row is not the cell's last
    if (!shared) { k = 1; mine = lane == 0 ? (int)r : -1; }   // too many rows for one wave: every row adds its own terms

    // lane i < k: target box, anchor and class of its row
    float tb0 = 0.f, tb1 = 0.f, tb2 = 0.f, tb3 = 0.f, an0 = 0.f, an1 = 0.f;
    int mcls = 0;
    if (lane < k) {
        if (mine == (int)r) {
            tb0 = R.tb[0]; tb1 = R.tb[1]; tb2 = R.tb[2]; tb3 = R.tb[3]; an0 = R.an[0]; an1 = R.an[1]; mcls = R.cls;
        } else {
            const float4 tb = *reinterpret_cast<const float4*>(tbox + ((size_t)l * P.cap + mine) * 4);
            const float2 an = *reinterpret_cast<const float2*>(anch + ((size_t)l * P.cap + mine) * 2);
            tb0 = tb.x; tb1 = tb.y; tb2 = tb.z; tb3 = tb.w; an0 = an.x; an1 = an.y;
            mcls = (int)tcls[(size_t)l * P.cap + mine];
        }
    }
    // order[t] (held by lane t): the lane whose row has the t-th smallest id
    int rank = 0, order = 0;
    for (int t = 0; t < k; ++t) rank += __shfl(mine, t, 64) < mine ? 1 : 0;
    for (int t = 0; t < k; ++t) order = __shfl(rank, t, 64) == lane ? t : order;

    float gbox = 0.0f;
    for (int t = 0; t < k; ++t) {
        const int src = __shfl(order, t, 64);
        const float b0 = __shfl(tb0, src, 64), b1 = __shfl(tb1, src, 64), b2 = __shfl(tb2, src, 64), b3 = __shfl(tb3, src, 64);
        const float a0 = __shfl(an0, src, 64), a1 = __shfl(an1, src, 64);
        if (lane < 4) gbox += box_term(lane, s0, s1, s2, s3, a0, a1, b0, b1, b2, b3, g_iou);      // in row order, like index_put(accumulate)
    }
    if (lane < 4) put_el<G>(gs, g0 + lane * cs, gbox, !shared);
    if (P.nc > 1) {
        for (int c0 = 0; c0 < P.nc; c0 += 64) {
            const int c = c0 + lane;
            const float x = c0 == 0 ? xc0 : (c0 == 64 ? xc1 : (c < P.nc ? ld_el<T>(ps, e0 + (5 + c) * cs) : 0.0f));
            float acc = 0.0f;
            for (int t = 0; t < k; ++t) {
                const int cls_t = __shfl(mcls, __shfl(order, t, 64), 64);
                acc += sc * bce_grad(x, c == cls_t ? P.cp : P.cn, P.cls_pw, P.gamma);
            }
            if (c < P.nc) put_el<G>(gs, g0 + (5 + c) * cs, acc, !shared);
        }
    }
}

struct yl_ws {
    size_t off_owner, off_iou, off_cls, off_partial, off_invalid, off_dobj, off_next, off_sums, total;
    long cells_total;
    unsigned obj_blocks;
};

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

bool make_params(const float* const* p, float* const* g, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                 long cap, const dh_yolo_hyp* hyp, bool grad_blocks, yl_params& P, yl_ws& W, bool conv = false)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || bs < 1 || na < 1 || nc < 1 || cap < 0 || !ny || !nx) return false;
    P.nl = nl; P.bs = bs; P.na = na; P.nc = nc; P.no = nc + 5; P.cap = cap;
    P.conv = conv ? 1 : 0;
    P.half = 0;
    P.half_grad = 0;
    P.level_scale = nullptr;
    if (hyp) {
        P.box = hyp->box; P.obj = hyp->obj; P.cls = hyp->cls; P.cls_pw = hyp->cls_pw; P.obj_pw = hyp->obj_pw;
        P.gamma = hyp->fl_gamma; P.cp = hyp->cp; P.cn = hyp->cn; P.gr = hyp->gr;
        P.sort_iou = hyp->sort_obj_iou != 0;
    }
    long first = 0;
    unsigned blocks = 0;
    uint32_t row_first = 0;
    for (int l = 0; l < nl; ++l) {
        if (ny[l] < 1 || nx[l] < 1) return false;
        yl_level& L = P.lvl[l];
        L.per_image = dh_make_fastdiv((uint32_t)(na * ny[l] * nx[l]));
        L.plane = dh_make_fastdiv((uint32_t)(ny[l] * nx[l]));
        L.gchunks = dh_make_fastdiv(1);
        L.row_first = row_first;
        row_first += (uint32_t)(na * ny[l] * nx[l]);
        L.p = p ? p[l] : nullptr;
        L.g = g ? g[l] : nullptr;
        L.ny = ny[l]; L.nx = nx[l];
        L.cells = (long)bs * na * ny[l] * nx[l];
        if ((unsigned long long)L.cells * P.no >= (1ull << 32)) return false;
        L.cell_first = first;
        first += L.cells;
        L.fwd_blocks = (unsigned)((L.cells + OBJ_CELLS_PER_BLOCK - 1) / OBJ_CELLS_PER_BLOCK);
        if (grad_blocks) {
            const size_t total = (size_t)L.cells * P.no;
            const size_t nvec = (total % 4 == 0) ? total / 4 : total;
            L.blocks = (unsigned)((nvec + GRAD_VEC_PER_BLOCK - 1) / GRAD_VEC_PER_BLOCK);
        } else {
            L.blocks = (unsigned)((L.cells + OBJ_CELLS_PER_BLOCK - 1) / OBJ_CELLS_PER_BLOCK);
        }
        blocks += L.blocks;
        L.balance = hyp ? hyp->balance[l] : 0.0f;
    }
    P.rows_per_image = row_first;
    W.cells_total = first;
    // workspace layout is independent of grad_blocks: obj_blocks is always the forward count
    unsigned fwd_blocks = 0;
    for (int l = 0; l < nl; ++l) fwd_blocks += (unsigned)((P.lvl[l].cells + OBJ_CELLS_PER_BLOCK - 1) / OBJ_CELLS_PER_BLOCK);
    W.obj_blocks = fwd_blocks;
    size_t o = 0;
    W.off_owner = o; o = up256(o + sizeof(int) * ((size_t)first + 1));   // + the ignored-row counter, cleared by the same fill
    W.off_iou = o; o = up256(o + sizeof(float) * (size_t)nl * (cap > 0 ? cap : 1));
    W.off_cls = o; o = up256(o + sizeof(float) * (size_t)nl * (cap > 0 ? cap : 1));
    W.off_partial = o; o = up256(o + sizeof(double) * fwd_blocks);
    W.off_dobj = o; o = up256(o + sizeof(float) * (size_t)first);
    W.off_next = o; o = up256(o + sizeof(int) * (size_t)nl * (cap > 0 ? cap : 1));      // per row: the row before it in its cell's list
    W.off_sums = o; o = up256(o + sizeof(double) * 5 * (size_t)nl);        // per level: sum(1-iou), sum class BCE, sum objectness BCE, rows, cells
    W.off_invalid = W.off_owner + sizeof(int) * (size_t)first;
    W.total = o;
    (void)blocks;
    return true;
}

unsigned sum_blocks(const yl_params& P)
{
    unsigned b = 0;
    for (int l = 0; l < P.nl; ++l) b += P.lvl[l].blocks;
    return b;
}

}  // namespace

extern "C" size_t dh_yolo_loss_workspace(const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t cap)
{
    yl_params P;
    yl_ws W;
    if (!make_params(nullptr, nullptr, ny, nx, nl, bs, na, nc, (long)cap, nullptr, false, P, W)) return 0;
    return W.total;
}

extern "C" size_t dh_yolo_loss_sums_offset(const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t cap)
{
    yl_params P;
    yl_ws W;
    if (!make_params(nullptr, nullptr, ny, nx, nl, bs, na, nc, (long)cap, nullptr, false, P, W)) return 0;
    return W.off_sums;
}

namespace {

// forward launches; fill: set the per-cell map (and the two counters behind it) to -1 here, false when the caller's
// assignment launch already did; final: run the reduction as a launch of its own
// parts: 1 = positives, 2 = dense objectness (+ final), 3 = both
int launch_fwd(const yl_params& P, const yl_ws& W, char* ws, const int64_t* idx, const float* tbox, const float* anch,
               const int64_t* tcls, const int* counts, const float* obj_logits, float* out, bool fill, bool final, hipStream_t st,
               int parts = 3)
{
    int* owner = (int*)(ws + W.off_owner);
    float* iou_ws = (float*)(ws + W.off_iou);
    float* cls_ws = (float*)(ws + W.off_cls);
    double* partial = (double*)(ws + W.off_partial);
    int* invalid = (int*)(ws + W.off_invalid);
    int* nxt = (int*)(ws + W.off_next);
    // one fill: per-cell map = -1 (no row), ignored-row counter = -1 (so the count is counter + 1)
    // the fill covers the region up to its 256-byte padding: a size that is a multiple of 16 is one fill kernel, not two
    if (fill && dh_fill_async(owner, 0xFF, W.off_iou - W.off_owner, st) != DH_OK) return DH_ERR_LAUNCH;
    if (P.cap > 0 && (parts & 1)) {
        if (P.conv && P.half) hipLaunchKernelGGL((k_yl_pos_fwd<true, _Float16>), dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P,
                                                 (const long long*)idx, tbox, anch, (const long long*)tcls, counts, owner, nxt, iou_ws, cls_ws, invalid);
        else if (P.conv) hipLaunchKernelGGL(k_yl_pos_fwd<true>, dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P,
                                       (const long long*)idx, tbox, anch, (const long long*)tcls, counts, owner, nxt, iou_ws, cls_ws, invalid);
        else hipLaunchKernelGGL(k_yl_pos_fwd<false>, dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P,
                                (const long long*)idx, tbox, anch, (const long long*)tcls, counts, owner, nxt, iou_ws, cls_ws, invalid);
    }
    if (!(parts & 2)) return dh_launch_status();
    // 16-byte vectors when the compact logits are there and every level keeps four consecutive cells inside one image
    bool vec = obj_logits != nullptr && dh_aligned16(obj_logits) && P.rows_per_image % 4 == 0;
    for (int l = 0; l < P.nl; ++l)
        vec = vec && P.lvl[l].per_image.d % 4 == 0 && P.lvl[l].row_first % 4 == 0 && P.lvl[l].cell_first % 4 == 0;
    if (P.conv && P.half) hipLaunchKernelGGL((k_yl_obj_fwd<false, true, _Float16>), dim3(sum_blocks(P)), dim3(256), 0, st, P, owner, nxt, iou_ws,
                                             partial, (float*)(ws + W.off_dobj), obj_logits);
    else if (P.conv) hipLaunchKernelGGL((k_yl_obj_fwd<false, true>), dim3(sum_blocks(P)), dim3(256), 0, st, P, owner, nxt, iou_ws, partial,
                                   (float*)(ws + W.off_dobj), obj_logits);
    else if (vec) hipLaunchKernelGGL(k_yl_obj_fwd<true>, dim3(sum_blocks(P)), dim3(256), 0, st, P, owner, nxt, iou_ws, partial,
                                (float*)(ws + W.off_dobj), obj_logits);
    else hipLaunchKernelGGL(k_yl_obj_fwd<false>, dim3(sum_blocks(P)), dim3(256), 0, st, P, owner, nxt, iou_ws, partial,
                            (float*)(ws + W.off_dobj), obj_logits);
    if (final)
        hipLaunchKernelGGL(k_yl_final, dim3(1), dim3(FINAL_THREADS), 0, st, P, counts, iou_ws, cls_ws, partial, invalid, out,
                           (double*)(ws + W.off_sums));
    return dh_launch_status();
}

// backward launches; F != nullptr: workgroup 0 of the dense pass also reduces the forward's partial sums into F->out
int launch_bwd(yl_params& P, const yl_ws& W, const char* ws, const int64_t* idx, const float* tbox, const float* anch,
               const int64_t* tcls, const int* counts, const float* grad_loss, float* const* grad_p, const yl_final_args* F,
               hipStream_t st)
{
    const int* owner = (const int*)(ws + W.off_owner);
    const int* nxt = (const int*)(ws + W.off_next);
    if (P.conv) {
        bool v4 = true;
        for (int l = 0; l < P.nl; ++l) v4 = v4 && P.lvl[l].plane.d % 4 == 0 && dh_aligned16(grad_p[l]);
        v4 = v4 && ((reinterpret_cast<uintptr_t>(ws) + W.off_dobj) & 15u) == 0;
        for (int l = 0; l < P.nl; ++l) {
            const uint32_t span = (uint32_t)P.no * P.lvl[l].plane.d;
            const uint32_t chunks = (span / (v4 ? 4 : 1) + 255) / 256;
            P.lvl[l].gchunks = dh_make_fastdiv(chunks);
            P.lvl[l].blocks = (unsigned)((size_t)P.bs * P.na * chunks);
        }
        const float* dobj_c = (const float*)(ws + W.off_dobj);
        const yl_final_args none_c = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const dim3 grid_c(sum_blocks(P));
        if (P.half_grad) {
            if (F) {
                if (v4) hipLaunchKernelGGL((k_yl_obj_bwd_conv<4, true, _Float16>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, *F);
                else hipLaunchKernelGGL((k_yl_obj_bwd_conv<1, true, _Float16>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, *F);
            } else {
                if (v4) hipLaunchKernelGGL((k_yl_obj_bwd_conv<4, false, _Float16>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, none_c);
                else hipLaunchKernelGGL((k_yl_obj_bwd_conv<1, false, _Float16>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, none_c);
            }
            if (P.cap > 0)
                hipLaunchKernelGGL((k_yl_pos_bwd<true, _Float16, _Float16>), dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P,
                                   (const long long*)idx, tbox, anch, (const long long*)tcls, counts, grad_loss, owner, nxt);
            return dh_launch_status();
        }
        if (F) {
            if (v4) hipLaunchKernelGGL((k_yl_obj_bwd_conv<4, true>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, *F);
            else hipLaunchKernelGGL((k_yl_obj_bwd_conv<1, true>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, *F);
        } else {
            if (v4) hipLaunchKernelGGL((k_yl_obj_bwd_conv<4, false>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, none_c);
            else hipLaunchKernelGGL((k_yl_obj_bwd_conv<1, false>), grid_c, dim3(256), 0, st, P, dobj_c, grad_loss, none_c);
        }
        if (P.cap > 0 && P.half)
            hipLaunchKernelGGL((k_yl_pos_bwd<true, _Float16>), dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P,
                               (const long long*)idx, tbox, anch, (const long long*)tcls, counts, grad_loss, owner, nxt);
        else if (P.cap > 0)
            hipLaunchKernelGGL(k_yl_pos_bwd<true>, dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P, (const long long*)idx,
                               tbox, anch, (const long long*)tcls, counts, grad_loss, owner, nxt);
        return dh_launch_status();
    }
    bool vec4 = true;
    for (int l = 0; l < P.nl; ++l) vec4 = vec4 && ((size_t)P.lvl[l].cells * P.no) % 4 == 0 && dh_aligned16(grad_p[l]);
    if (!vec4) {   // scalar stores: one element per "vector"
        for (int l = 0; l < P.nl; ++l) {
            const size_t total = (size_t)P.lvl[l].cells * P.no;
            P.lvl[l].blocks = (unsigned)((total + GRAD_VEC_PER_BLOCK - 1) / GRAD_VEC_PER_BLOCK);
        }
    }
    const float* dobj = (const float*)(ws + W.off_dobj);
    const dh_fastdiv nod = dh_make_fastdiv((uint32_t)P.no);
    const yl_final_args none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid(sum_blocks(P));
    if (F) {
        if (vec4) hipLaunchKernelGGL((k_yl_obj_bwd<4, true>), grid, dim3(256), 0, st, P, dobj, grad_loss, nod, *F);
        else hipLaunchKernelGGL((k_yl_obj_bwd<1, true>), grid, dim3(256), 0, st, P, dobj, grad_loss, nod, *F);
    } else {
        if (vec4) hipLaunchKernelGGL((k_yl_obj_bwd<4, false>), grid, dim3(256), 0, st, P, dobj, grad_loss, nod, none);
        else hipLaunchKernelGGL((k_yl_obj_bwd<1, false>), grid, dim3(256), 0, st, P, dobj, grad_loss, nod, none);
    }
    if (P.cap > 0)
        hipLaunchKernelGGL(k_yl_pos_bwd<false>, dim3((unsigned)((P.cap + 3) / 4), P.nl), dim3(256), 0, st, P, (const long long*)idx, tbox,
                           anch, (const long long*)tcls, counts, grad_loss, owner, nxt);
    return dh_launch_status();
}

}  // namespace

static int loss_fwd(int conv, const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                    const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                    const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* obj_logits, float* out,
                    void* workspace, size_t workspace_bytes, void* stream)
{
    yl_params P;
    yl_ws W;
    if (!p || !hyp || !out || !counts || !workspace) return DH_ERR_ARG;
    if (!make_params(p, nullptr, ny, nx, nl, bs, na, nc, (long)cap, hyp, false, P, W, conv != 0)) return DH_ERR_ARG;
    P.half = conv >= 2;
    for (int l = 0; l < nl; ++l) if (!p[l]) return DH_ERR_ARG;
    if (cap > 0 && (!idx || !tbox || !anch || !tcls)) return DH_ERR_ARG;
    if (workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) || (cap > 0 && (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u))))
        return DH_ERR_ALIGN;
    return launch_fwd(P, W, (char*)workspace, idx, tbox, anch, tcls, counts, obj_logits, out, true, true, (hipStream_t)stream);
}

extern "C" int dh_yolo_loss_fwd(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* obj_logits, float* out,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    return loss_fwd(0, p, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp, obj_logits, out, workspace,
                    workspace_bytes, stream);
}

extern "C" int dh_yolo_loss_fwd_conv(const float* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                     const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                     const int* counts, int64_t cap, const dh_yolo_hyp* hyp, float* out, void* workspace,
                                     size_t workspace_bytes, void* stream)
{
    return loss_fwd(1, conv, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp, nullptr, out, workspace,
                    workspace_bytes, stream);
}

static int loss_bwd(int conv, const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                    const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                    const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss,
                    float* const* grad_p, const void* workspace, size_t workspace_bytes, void* stream,
                    const float* level_scale = nullptr)
{
    yl_params P;
    yl_ws W;
    if (!p || !grad_p || !hyp || !grad_loss || !counts || !workspace) return DH_ERR_ARG;
    if (!make_params(p, grad_p, ny, nx, nl, bs, na, nc, (long)cap, hyp, true, P, W, conv != 0)) return DH_ERR_ARG;
    P.half = conv >= 2;
    P.half_grad = conv == 3;
    P.level_scale = level_scale;
    for (int l = 0; l < nl; ++l) if (!p[l] || !grad_p[l]) return DH_ERR_ARG;
    if (cap > 0 && (!idx || !tbox || !anch || !tcls)) return DH_ERR_ARG;
    if (workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    // the mate rows of a shared cell are read as 16-byte / 8-byte vectors
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) || (cap > 0 && (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u))))
        return DH_ERR_ALIGN;
    return launch_bwd(P, W, (const char*)workspace, idx, tbox, anch, tcls, counts, grad_loss, grad_p, nullptr, (hipStream_t)stream);
}

extern "C" int dh_yolo_loss_bwd(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss,
                                float* const* grad_p, const void* workspace, size_t workspace_bytes, void* stream)
{
    return loss_bwd(0, p, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp, grad_loss, grad_p, workspace,
                    workspace_bytes, stream);
}

extern "C" int dh_yolo_loss_bwd_conv(const float* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                     const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                     const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss,
                                     float* const* grad_conv, const void* workspace, size_t workspace_bytes, void* stream)
{
    return loss_bwd(1, conv, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp, grad_loss, grad_conv, workspace,
                    workspace_bytes, stream);
}

static int loss_fwd_bwd(int conv, const float* targets, int64_t nt, const float* anchors, double anchor_t,
                        const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                        int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                        const dh_yolo_hyp* hyp, const float* obj_logits, const float* grad_loss,
                        float* const* grad_p, float* out, void* assign_workspace, size_t assign_workspace_bytes,
                        void* workspace, size_t workspace_bytes, void* stream)
{
    yl_params P, Pb;
    yl_ws W, Wb;
    if (!p || !grad_p || !hyp || !out || !grad_loss || !counts || !workspace) return DH_ERR_ARG;
    if (!make_params(p, nullptr, ny, nx, nl, bs, na, nc, (long)cap, hyp, false, P, W, conv != 0)) return DH_ERR_ARG;
    if (!make_params(p, grad_p, ny, nx, nl, bs, na, nc, (long)cap, hyp, true, Pb, Wb, conv != 0)) return DH_ERR_ARG;
    P.half = Pb.half = conv >= 2;
    Pb.half_grad = conv == 3;
    for (int l = 0; l < nl; ++l) if (!p[l] || !grad_p[l]) return DH_ERR_ARG;
    if (cap > 0 && (!idx || !tbox || !anch || !tcls)) return DH_ERR_ARG;
    if (workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) || (cap > 0 && (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u))))
        return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    // 1 assignment; spare workgroups of that launch set the per-cell map and the counters behind it to -1
    int rc = dh_launch_build_targets(targets, nt, anchors, nl, na, ny, nx, anchor_t, cap, idx, tbox, anch, tcls, counts,
                                     assign_workspace, assign_workspace_bytes, ws + W.off_owner, W.off_iou - W.off_owner, stream);
    if (rc != DH_OK) return rc;
    // 2 positives, 3 dense objectness (no reduction launch)
    rc = launch_fwd(P, W, ws, idx, tbox, anch, tcls, counts, obj_logits, out, false, false, (hipStream_t)stream);
    if (rc != DH_OK) return rc;
    // 4 dense gradient, whose workgroup 0 also does the forward's reduction, 5 positives' gradient
    const yl_final_args F = {counts, (const float*)(ws + W.off_iou), (const float*)(ws + W.off_cls),
                             (const double*)(ws + W.off_partial), (const int*)(ws + W.off_invalid), out, (double*)(ws + W.off_sums)};
    return launch_bwd(Pb, Wb, ws, idx, tbox, anch, tcls, counts, grad_loss, grad_p, &F, (hipStream_t)stream);
}

extern "C" int dh_yolo_loss_fwd_bwd(const float* targets, int64_t nt, const float* anchors, double anchor_t,
                                    const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                    int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                                    const dh_yolo_hyp* hyp, const float* obj_logits, const float* grad_loss,
                                    float* const* grad_p, float* out, void* assign_workspace, size_t assign_workspace_bytes,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    return loss_fwd_bwd(0, targets, nt, anchors, anchor_t, p, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp,
                        obj_logits, grad_loss, grad_p, out, assign_workspace, assign_workspace_bytes, workspace, workspace_bytes,
                        stream);
}

extern "C" int dh_yolo_loss_fwd_bwd_conv(const float* targets, int64_t nt, const float* anchors, double anchor_t,
                                         const float* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                         int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                                         const dh_yolo_hyp* hyp, const float* grad_loss, float* const* grad_conv, float* out,
                                         void* assign_workspace, size_t assign_workspace_bytes, void* workspace,
                                         size_t workspace_bytes, void* stream)
{
    return loss_fwd_bwd(1, targets, nt, anchors, anchor_t, conv, ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap,
                        hyp, nullptr, grad_loss, grad_conv, out, assign_workspace, assign_workspace_bytes, workspace,
                        workspace_bytes, stream);
}

extern "C" int dh_yolo_loss_fwd_conv_f16(const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                         const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                         const int* counts, int64_t cap, const dh_yolo_hyp* hyp, float* out, void* workspace,
                                         size_t workspace_bytes, void* stream)
{
    return loss_fwd(2, reinterpret_cast<const float* const*>(conv), ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp,
                    nullptr, out, workspace, workspace_bytes, stream);
}

extern "C" int dh_yolo_loss_bwd_conv_f16(const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                         const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                         const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss,
                                         int grad_f16, void* const* grad_conv, const void* workspace, size_t workspace_bytes,
                                         void* stream)
{
    return loss_bwd(grad_f16 ? 3 : 2, reinterpret_cast<const float* const*>(conv), ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp,
                    grad_loss, reinterpret_cast<float* const*>(grad_conv), workspace, workspace_bytes, stream);
}

extern "C" int dh_yolo_loss_fwd_bwd_conv_f16(const float* targets, int64_t nt, const float* anchors, double anchor_t,
                                             const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                             int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                                             const dh_yolo_hyp* hyp, const float* grad_loss, int grad_f16, void* const* grad_conv,
                                             float* out, void* assign_workspace, size_t assign_workspace_bytes, void* workspace,
                                             size_t workspace_bytes, void* stream)
{
    return loss_fwd_bwd(grad_f16 ? 3 : 2, targets, nt, anchors, anchor_t, reinterpret_cast<const float* const*>(conv), ny, nx, nl, bs, na, nc, idx,
                        tbox, anch, tcls, counts, cap, hyp, nullptr, grad_loss, reinterpret_cast<float* const*>(grad_conv), out,
                        assign_workspace, assign_workspace_bytes, workspace, workspace_bytes, stream);
}

// The fused entry in two halves, for callers that can start the first one early: assignment and the positives' forward
// only read the targets and the raw maps, so they can be enqueued on another stream beside whatever produces or decodes
// the maps; the second half (dense objectness forward, the whole backward, the reduction) follows on the stream that
// holds the objectness logits.  Same arguments and workspaces as dh_yolo_loss_fwd_bwd; the caller orders the two calls
// (an event between the streams).
extern "C" int dh_yolo_loss_positives(const float* targets, int64_t nt, const float* anchors, double anchor_t,
                                      const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                      int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                                      const dh_yolo_hyp* hyp, void* assign_workspace, size_t assign_workspace_bytes,
                                      void* workspace, size_t workspace_bytes, void* stream)
{
    yl_params P;
    yl_ws W;
    if (!p || !hyp || !counts || !workspace) return DH_ERR_ARG;
    if (!make_params(p, nullptr, ny, nx, nl, bs, na, nc, (long)cap, hyp, false, P, W)) return DH_ERR_ARG;
    for (int l = 0; l < nl; ++l) if (!p[l]) return DH_ERR_ARG;
    if (cap > 0 && (!idx || !tbox || !anch || !tcls)) return DH_ERR_ARG;
    if (workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) || (cap > 0 && (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u))))
        return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    const int rc = dh_launch_build_targets(targets, nt, anchors, nl, na, ny, nx, anchor_t, cap, idx, tbox, anch, tcls, counts,
                                           assign_workspace, assign_workspace_bytes, ws + W.off_owner, W.off_iou - W.off_owner, stream);
    if (rc != DH_OK) return rc;
    return launch_fwd(P, W, ws, idx, tbox, anch, tcls, counts, nullptr, nullptr, false, false, (hipStream_t)stream, 1);
}

extern "C" int dh_yolo_loss_dense(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                  const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                                  int64_t cap, const dh_yolo_hyp* hyp, const float* obj_logits, const float* grad_loss,
                                  float* const* grad_p, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    yl_params P, Pb;
    yl_ws W, Wb;
    if (!p || !grad_p || !hyp || !out || !grad_loss || !counts || !workspace) return DH_ERR_ARG;
    if (!make_params(p, nullptr, ny, nx, nl, bs, na, nc, (long)cap, hyp, false, P, W)) return DH_ERR_ARG;
    if (!make_params(p, grad_p, ny, nx, nl, bs, na, nc, (long)cap, hyp, true, Pb, Wb)) return DH_ERR_ARG;
    for (int l = 0; l < nl; ++l) if (!p[l] || !grad_p[l]) return DH_ERR_ARG;
    if (cap > 0 && (!idx || !tbox || !anch || !tcls)) return DH_ERR_ARG;
    if (workspace_bytes < W.total) return DH_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) || (cap > 0 && (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u))))
        return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    const int rc = launch_fwd(P, W, ws, idx, tbox, anch, tcls, counts, obj_logits, out, false, false, (hipStream_t)stream, 2);
    if (rc != DH_OK) return rc;
    const yl_final_args F = {counts, (const float*)(ws + W.off_iou), (const float*)(ws + W.off_cls),
                             (const double*)(ws + W.off_partial), (const int*)(ws + W.off_invalid), out, (double*)(ws + W.off_sums)};
    return launch_bwd(Pb, Wb, ws, idx, tbox, anch, tcls, counts, grad_loss, grad_p, &F, (hipStream_t)stream);
}

// The backward of any of the layouts with a factor per level on top of the upstream gradient: level_scale[2*l] multiplies the
// box and class terms of level l, level_scale[2*l + 1] its objectness term (device memory, nl pairs).  This is what turns the
// per-rank normalisation of the reference (object_detection/yolov5/utils/loss.py:136,149,155: means over the rank's own rows
// and cells) into the normalisation of the whole data-parallel batch: factor = world * local divisor / global divisor, with
// the global divisors taken from the all-reduced sums the forward leaves at dh_yolo_loss_sums_offset.
// layout: 0 maps (bs,na,ny,nx,no) fp32; 1 convolution layout fp32; 2 convolution layout fp16, fp32 gradient; 3 fp16 gradient.
extern "C" int dh_yolo_loss_bwd_scaled(int layout, const void* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                       const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls,
                                       const int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss,
                                       const float* level_scale, void* const* grad_p, const void* workspace,
                                       size_t workspace_bytes, void* stream)
{
    if (layout < 0 || layout > 3 || !level_scale) return DH_ERR_ARG;
    return loss_bwd(layout, reinterpret_cast<const float* const*>(p), ny, nx, nl, bs, na, nc, idx, tbox, anch, tcls, counts, cap, hyp,
                    grad_loss, reinterpret_cast<float* const*>(grad_p), workspace, workspace_bytes, stream, level_scale);
}
