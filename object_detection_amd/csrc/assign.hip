// YOLOv5 target assignment (anchor matching + neighbour-cell expansion) for all levels in one launch.
//
// Reference: object_detection/yolov5/utils/loss.py:169-222 (ComputeLoss.build_targets).
//   for each level: t = targets * [1,1,nx,ny,nx,ny]; keep (anchor a, target t) when
//   max(r, 1/r) over w,h < anchor_t with r = gwh / anchors[a]; expand each kept row to the cell
//   itself plus up to four neighbours chosen by the fractional position of gxy; cell indices are
//   trunc(gxy - offset) clamped to the grid; tbox = (gxy - cell, gwh).
// Output order is the reference's: offset-major {centre,+x,+y,-x,-y}, then anchor-major, then target
// order -- two row-major boolean compactions (loss.py:194,202).  It is reproduced with
// order-preserving ballot/prefix-scan compaction (no atomics), one workgroup per level.
// The clamp of loss.py:217 works in place on views of the cell tensor, so tbox uses clamped cells.
#include "dh_common.h"
#include "dh_internal.h"
#include "../../include/dethead.h"

namespace {

#define BT_THREADS 1024
#define BT_CHUNKS 2       // rows per thread that the single-sweep expansion keeps in registers
#define BT_LDS_T 1024     // targets kept in LDS (24 KB): every later read of a target is then an LDS read
#define BT_LDS_Q 4096     // (anchor, target) pairs whose stage-1 list stays in LDS instead of going through global memory

struct bt_params {
    int ny[DH_MAX_LEVELS];
    int nx[DH_MAX_LEVELS];
    int nl, na;
    int nt;
    dh_fastdiv ntd;   // nt
    float anchor_t;
    long cap;   // rows reserved per level = 5*na*nt
};

// Position of this thread's flag among the set flags of the workgroup (row-major) and their total.  Consecutive
// calls alternate between two count buffers (`phase` toggles), so one barrier per call is enough: nobody can still
// be reading buffer k when it is written again two calls later, because the call in between has its own barrier.
__device__ __forceinline__ int block_rank(bool flag, int& total, int (*s_w)[BT_THREADS / 64], int& phase)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* cnt = s_w[phase];
    phase ^= 1;
    if (lane == 0) cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < BT_THREADS / 64; ++w) {
        const int c = cnt[w];
        if (w < wave) before += c;
        all += c;
    }
    total = all;
    return before + __popcll(m & ((1ull << lane) - 1ull));
}

// torch.remainder(x, 1) for fp32: fmod(x, 1), then a negative result is shifted up by the divisor.  fmod(x, 1) is
// x - trunc(x), which fp32 forms exactly (the fraction bits of x), so no general fmod is needed: the kernel is one
// workgroup per level and bound by the instructions its 16 waves issue on one compute unit.
__device__ __forceinline__ float remainder_one(float x)
{
    float m = x - truncf(x);
    if (m != 0.0f && m < 0.0f) m += 1.0f;
    return m;
}

__global__ void __launch_bounds__(BT_THREADS) k_build_targets(const float* targets,
                                                              const float* __restrict__ anchors, const bt_params P,
                                                              int* __restrict__ list_ws, long long* __restrict__ idx,
                                                              float* __restrict__ tbox, float* __restrict__ anch,
                                                              long long* __restrict__ tcls, int* __restrict__ counts,
                                                              uint4* __restrict__ fill, size_t fill_vecs)
{
    __shared__ int s_w[2][BT_THREADS / 64];
    int phase = 0;
    if ((int)blockIdx.x >= 5 * P.nl) {  // side duty of the same launch: set a caller's region to all ones (16 B per store)
        const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        const size_t step = (size_t)(gridDim.x - 5 * P.nl) * BT_THREADS;
        for (size_t i = (size_t)(blockIdx.x - 5 * P.nl) * BT_THREADS + threadIdx.x; i < fill_vecs; i += step) fill[i] = ones;
        return;
    }
    // Five workgroups per level, one per neighbour offset: each repeats the (cheap) ratio test and the counting of all five
    // offsets, which tells it where its own offset's rows start, and writes only those rows -- the row stores of a level
    // then leave from five compute units instead of one.
    const int l = blockIdx.x / 5, my_off = blockIdx.x - 5 * l;
    const int nx = P.nx[l], ny = P.ny[l];
    const float fnx = (float)nx, fny = (float)ny;
    const int total_q = P.na * P.nt;
    // The kernel is one workgroup per level and nothing but dependent steps: what it reads more than once comes from LDS
    // when it fits (the usual case), so that the only global round trips are the first read of the targets and the stores.
    __shared__ float s_targets[BT_LDS_T * 6];
    __shared__ int s_list[BT_LDS_Q];
    const float* targets_in = targets;
    if (P.nt <= BT_LDS_T) {
        for (int i = threadIdx.x; i < P.nt * 6; i += BT_THREADS) s_targets[i] = targets_in[i];
        __syncthreads();
        targets = s_targets;
    }
    int* list = total_q <= BT_LDS_Q ? s_list : list_ws + ((size_t)l * 5 + my_off) * total_q;
    const float* __restrict__ anc = anchors + (size_t)l * P.na * 2;

    // stage 1: (anchor, target) pairs that pass the ratio test, in anchor-major order (loss.py:191-194)
    int n1 = 0;
    for (int q0 = 0; q0 < total_q; q0 += BT_THREADS) {
        const int q = q0 + threadIdx.x;
        bool ok = false;
        if (q < total_q) {
            const int a = (int)dh_div((uint32_t)q, P.ntd), t = q - a * P.nt;
            const float gw = targets[(size_t)t * 6 + 4] * fnx, gh = targets[(size_t)t * 6 + 5] * fny;
            const float rw = __fdiv_rn(gw, anc[2 * a]), rh = __fdiv_rn(gh, anc[2 * a + 1]);
            const float worst = fmaxf(fmaxf(rw, __fdiv_rn(1.0f, rw)), fmaxf(rh, __fdiv_rn(1.0f, rh)));
            ok = worst < P.anchor_t;
        }
        int tot;
        const int rank = block_rank(ok, tot, s_w, phase);
        if (ok) list[n1 + rank] = q;
        n1 += tot;
    }
    __syncthreads();   // list[] written by this workgroup is read back below

    // stage 2: expansion to the neighbour cells, in offset-major order (loss.py:197-203)
    long long* __restrict__ ob = idx + (size_t)l * 4 * P.cap;
    long long* __restrict__ oa = ob + P.cap;
    long long* __restrict__ oj = oa + P.cap;
    long long* __restrict__ oi = oj + P.cap;
    float* __restrict__ obox = tbox + (size_t)l * P.cap * 4;
    float* __restrict__ oanch = anch + (size_t)l * P.cap * 2;
    long long* __restrict__ ocls = tcls + (size_t)l * P.cap;
    int n2 = 0;
    // Which of the five offsets a row takes (bit o of the result) -- loss.py:197-203
    auto offsets_taken = [&](float gx, float gy) -> unsigned {
        const float ix = fnx - gx, iy = fny - gy;
        unsigned bits = 1u;
        if ((remainder_one(gx) < 0.5f) && (gx > 1.0f)) bits |= 2u;
        if ((remainder_one(gy) < 0.5f) && (gy > 1.0f)) bits |= 4u;
        if ((remainder_one(ix) < 0.5f) && (ix > 1.0f)) bits |= 8u;
        if ((remainder_one(iy) < 0.5f) && (iy > 1.0f)) bits |= 16u;
        return bits;
    };
    auto emit = [&](size_t pos, int off, int a, int t, float gx, float gy) {
        const float ox = off == 1 ? 0.5f : (off == 3 ? -0.5f : 0.0f);
        const float oy = off == 2 ? 0.5f : (off == 4 ? -0.5f : 0.0f);
        long long gi = (long long)(gx - ox), gj = (long long)(gy - oy);           // .long(): truncation
        gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);                            // loss.py:217
        gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);
        const float* __restrict__ tr = targets + (size_t)t * 6;
        ob[pos] = (long long)tr[0];
        oa[pos] = a;
        oj[pos] = gj;
        oi[pos] = gi;
        // one 16-byte and one 8-byte store (the row stores of a level all leave from one compute unit: their number is
        // what the kernel's duration is made of)
        *reinterpret_cast<float4*>(obox + 4 * pos) = make_float4(gx - (float)gi, gy - (float)gj, tr[4] * fnx, tr[5] * fny);
        *reinterpret_cast<float2*>(oanch + 2 * pos) = make_float2(anc[2 * a], anc[2 * a + 1]);
        ocls[pos] = (long long)tr[1];
    };
    if (n1 <= BT_CHUNKS * BT_THREADS) {
        // The usual size: every thread owns at most BT_CHUNKS rows.  All offsets are counted in one sweep
        // (per offset, chunk and wave), one wave turns the counts into offset-major start positions, and a second
        // sweep writes -- three barriers instead of one per (offset, chunk).
        __shared__ int s_cnt[5 * BT_CHUNKS * (BT_THREADS / 64) + 1];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned bits[BT_CHUNKS];
        int ra[BT_CHUNKS], rt[BT_CHUNKS];
        float rgx[BT_CHUNKS], rgy[BT_CHUNKS];
#pragma unroll
        for (int ch = 0; ch < BT_CHUNKS; ++ch) {
            const int k = ch * BT_THREADS + threadIdx.x;
            bits[ch] = 0; ra[ch] = 0; rt[ch] = 0; rgx[ch] = 0.f; rgy[ch] = 0.f;
            if (k < n1) {
                const int q = list[k];
                ra[ch] = (int)dh_div((uint32_t)q, P.ntd); rt[ch] = q - ra[ch] * P.nt;
                rgx[ch] = targets[(size_t)rt[ch] * 6 + 2] * fnx;
                rgy[ch] = targets[(size_t)rt[ch] * 6 + 3] * fny;
                bits[ch] = offsets_taken(rgx[ch], rgy[ch]);
            }
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                const unsigned long long m = __ballot((bits[ch] >> o) & 1u);
                if (lane == 0) s_cnt[(o * BT_CHUNKS + ch) * (BT_THREADS / 64) + wave] = __popcll(m);
            }
        }
        __syncthreads();
        if (wave == 0) {      // exclusive prefix sum over the 5 * BT_CHUNKS * 16 counts, in output order
            constexpr int N = 5 * BT_CHUNKS * (BT_THREADS / 64);
            constexpr int PER = (N + 63) / 64;
            int v[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) { const int i = lane * PER + j; v[j] = i < N ? s_cnt[i] : 0; sum += v[j]; }
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
            int run = incl - sum;
#pragma unroll
            for (int j = 0; j < PER; ++j) { const int i = lane * PER + j; if (i < N) s_cnt[i] = run; run += v[j]; }
            if (lane == 63) s_cnt[N] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < BT_CHUNKS; ++ch) {
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                if (o != my_off) continue;
                const bool take = (bits[ch] >> o) & 1u;
                const unsigned long long m = __ballot(take);
                if (take) {
                    const size_t pos = (size_t)s_cnt[(o * BT_CHUNKS + ch) * (BT_THREADS / 64) + wave] +
                                       __popcll(m & ((1ull << lane) - 1ull));
                    emit(pos, o, ra[ch], rt[ch], rgx[ch], rgy[ch]);
                }
            }
        }
        n2 = s_cnt[5 * BT_CHUNKS * (BT_THREADS / 64)];
    } else {
        for (int off = 0; off < 5; ++off) {      // any size: one pass per neighbour offset keeps the offset-major order
            for (int k0 = 0; k0 < n1; k0 += BT_THREADS) {
                const int k = k0 + threadIdx.x;
                bool take = false;
                int a = 0, t = 0;
                float gx = 0.f, gy = 0.f;
                if (k < n1) {
                    const int q = list[k];
                    a = (int)dh_div((uint32_t)q, P.ntd); t = q - a * P.nt;
                    gx = targets[(size_t)t * 6 + 2] * fnx;
                    gy = targets[(size_t)t * 6 + 3] * fny;
                    take = (offsets_taken(gx, gy) >> off) & 1u;
                }
                int tot;
                const int rank = block_rank(take, tot, s_w, phase);
                if (take && off == my_off) emit((size_t)n2 + rank, off, a, t, gx, gy);
                n2 += tot;
            }
        }
    }
    if (threadIdx.x == 0 && my_off == 0) counts[l] = n2;
}

}  // namespace

extern "C" size_t dh_build_targets_workspace(int nl, int na, int64_t nt)
{
    if (nl <= 0 || na <= 0 || nt <= 0) return 256;
    return (size_t)nl * 5 * na * nt * sizeof(int) + 256;      // the stage-1 list of every (level, offset) workgroup
}

// Shared by dh_build_targets and the fused loss entry: the assignment launch, which can also fill `fill_bytes`
// (a multiple of 16, 16-byte aligned) at `fill` with 0xFF in spare workgroups of the same launch.
int dh_launch_build_targets(const float* targets, int64_t nt, const float* anchors, int nl, int na, const int* ny, const int* nx,
                            double anchor_t, int64_t cap, int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts,
                            void* workspace, size_t workspace_bytes, void* fill, size_t fill_bytes, void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || nt < 0 || !ny || !nx || !counts || !anchors) return DH_ERR_ARG;
    if ((long long)na * nt >= (1ll << 30)) return DH_ERR_ARG;
    if (fill_bytes && (!fill || (fill_bytes & 15u) || !dh_aligned16(fill))) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (nt == 0) {
        if (fill_bytes && dh_fill_async(fill, 0xFF, fill_bytes, st) != DH_OK) return DH_ERR_LAUNCH;
        return dh_fill_async(counts, 0, sizeof(int) * nl, st);
    }
    if (!targets || !idx || !tbox || !anch || !tcls || !workspace || cap < 5ll * na * nt) return DH_ERR_ARG;
    if (!dh_aligned16(tbox) || (reinterpret_cast<uintptr_t>(anch) & 7u)) return DH_ERR_ALIGN;     // rows are stored as vectors
    if (workspace_bytes < dh_build_targets_workspace(nl, na, nt)) return DH_ERR_WORKSPACE;
    bt_params P;
    for (int l = 0; l < nl; ++l) {
        if (ny[l] < 1 || nx[l] < 1) return DH_ERR_ARG;
        P.ny[l] = ny[l];
        P.nx[l] = nx[l];
    }
    P.nl = nl; P.na = na; P.nt = (int)nt; P.ntd = dh_make_fastdiv((uint32_t)nt); P.anchor_t = (float)anchor_t; P.cap = (long)cap;
    const size_t fill_vecs = fill_bytes / 16;
    // 32 KiB of fill per spare workgroup (two stores per thread), at most 512 of them
    size_t spare = (fill_vecs + 2 * BT_THREADS - 1) / (2 * BT_THREADS);
    if (spare > 512) spare = 512;
    hipLaunchKernelGGL(k_build_targets, dim3(5 * nl + (unsigned)spare), dim3(BT_THREADS), 0, st, targets, anchors, P, (int*)workspace,
                       (long long*)idx, tbox, anch, (long long*)tcls, counts, (uint4*)fill, fill_vecs);
    return dh_launch_status();
}

extern "C" int dh_build_targets(const float* targets, int64_t nt, const float* anchors, int nl, int na, const int* ny,
                                const int* nx, double anchor_t, int64_t cap, int64_t* idx, float* tbox, float* anch,
                                int64_t* tcls, int* counts, void* workspace, size_t workspace_bytes, void* stream)
{
    return dh_launch_build_targets(targets, nt, anchors, nl, na, ny, nx, anchor_t, cap, idx, tbox, anch, tcls, counts, workspace,
                                   workspace_bytes, nullptr, 0, stream);
}
