// Per-class average precision and the precision / recall / F1 curves of an evaluation run, on the device.
//
// Reference (object_detection/yolov5/utils/metrics.py):
//   :21-86   ap_per_class   sort by confidence; per class with labels: cumulative tp / fp, recall and precision curves,
//                           the two curves sampled at 1000 confidences (np.interp), AP per IoU threshold
//   :89-114  compute_ap     sentinels, precision envelope, 101-point interpolation, np.trapz
// Everything after the sort is float64 in the reference; the kernels use the same IEEE operations in the same order
// (no contraction), including np.interp's segment choice on repeated abscissae and the order in which np.sum adds the
// 100 trapezoids, so the results are the reference's bit for bit when the confidences are all different.  Detections of
// one class with equal confidence keep their input order (the reference's unstable sort may order them either way).
//
// Layout: detections get 64-bit keys [class | ~order(conf)] and are sorted once (the radix sort below, stable); each class
// is then a contiguous run in descending confidence.  One workgroup per (class, IoU threshold) scans its run forwards
// (cumulative tp) and backwards (precision envelope) through workspace arrays, then interpolates.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

#define AP_THREADS 1024
#define AP_WAVES (AP_THREADS / 64)
#define AP_PX 1000          // metrics.py:44
#define AP_X 101            // metrics.py:108

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t conf_desc_bits(float f)
{
    const uint32_t b = __float_as_uint(f);
    const uint32_t asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);      // ascending in f
    return ~asc;
}

__device__ __forceinline__ float conf_of_key(u64 key)
{
    const uint32_t asc = ~(uint32_t)key;
    return __uint_as_float(asc ^ ((asc >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// ---- classes that have labels (np.unique(target_cls, return_counts=True), metrics.py:40) ---------------------------------
// Label classes repeat (tens of thousands of labels over a few dozen ids) and atomics on one global address serialise,
// so a workgroup counts its AP_HIST_PER_THREAD * 1024 labels in an LDS histogram (ids below AP_HIST_LDS) first.
#define AP_HIST_LDS 4096
#define AP_HIST_PER_THREAD 8

__global__ void __launch_bounds__(AP_THREADS) k_ap_hist(const double* __restrict__ target_cls, long m, int class_cap,
                                                        int* __restrict__ bins, int* __restrict__ meta)
{
    __shared__ int s_bins[AP_HIST_LDS];
    for (int i = threadIdx.x; i < AP_HIST_LDS; i += AP_THREADS) s_bins[i] = 0;
    __syncthreads();
    const long first = (long)blockIdx.x * (AP_THREADS * AP_HIST_PER_THREAD);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < AP_HIST_PER_THREAD; ++k) {
        const long i = first + k * AP_THREADS + threadIdx.x;
        if (i >= m) continue;
        const double v = target_cls[i];
        const int c = (int)v;
        if (v == (double)c && c >= 0 && c < class_cap) {
            if (c < AP_HIST_LDS) atomicAdd(&s_bins[c], 1);
            else atomicAdd(&bins[c], 1);
        } else {
            bad = true;                                                    // not an integer class id in range
        }
    }
    if (bad) atomicOr(&meta[1], 1);
    __syncthreads();
    const int lim = class_cap < AP_HIST_LDS ? class_cap : AP_HIST_LDS;
    for (int i = threadIdx.x; i < lim; i += AP_THREADS)
        if (s_bins[i]) atomicAdd(&bins[i], s_bins[i]);
}

__global__ void __launch_bounds__(AP_THREADS) k_ap_unique(const int* __restrict__ bins, int class_cap, int* __restrict__ unique,
                                                          int* __restrict__ nt, int* __restrict__ meta)
{
    __shared__ int s_w[AP_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int total = 0;
    for (int base = 0; base < class_cap; base += AP_THREADS) {
        const int c = base + threadIdx.x;
        const int cnt = c < class_cap ? bins[c] : 0;
        const u64 bm = __ballot(cnt > 0);
        if (lane == 0) s_w[wave] = __popcll(bm);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < AP_WAVES; ++w) { const int k = s_w[w]; if (w < wave) before += k; all += k; }
        __syncthreads();
        if (cnt > 0) {
            const int pos = total + before + __popcll(bm & ((1ull << lane) - 1ull));
            unique[pos] = c;
            nt[pos] = cnt;
        }
        total += all;
    }
    if (threadIdx.x == 0) meta[0] = total;
}

// ---- sort keys -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ap_keys(const float* __restrict__ conf, const float* __restrict__ pred_cls, long n,
                                                 int class_cap, u64* __restrict__ keys, int* __restrict__ vals)
{
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= n) return;
    const float c = pred_cls[i];
    const int ci = (int)c;
    const bool ok = c == (float)ci && ci >= 0 && ci < class_cap;
    keys[i] = ((u64)(ok ? (uint32_t)ci : (uint32_t)class_cap) << 32) | conf_desc_bits(conf[i]);    // class_cap: behind every class
    vals[i] = (int)i;
}

// first position whose key is >= `key`
__device__ long lower_bound_keys(const u64* __restrict__ keys, long n, u64 key)
{
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// np.interp's choice of segment: the last j with xp[j] <= x (callers have handled x < xp[0] and x > xp[len-1])
template <typename XP>
__device__ __forceinline__ long interp_segment(long len, double x, XP xp)
{
    long lo = 0, hi = len;
    while (lo < hi) {
        const long mid = lo + ((hi - lo) >> 1);
        if (x >= xp(mid)) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// one point of np.interp(x, xp, fp, left, right) for float64 arrays given as index functions
template <typename XP, typename FP>
__device__ __forceinline__ double interp_at(double x, long len, XP xp, FP fp, double left, double right)
{
    if (x < xp(0)) return left;
    if (x > xp(len - 1)) return right;
    const long j = interp_segment(len, x, xp);
    if (j == len - 1) return fp(j);
    const double xj = xp(j), fj = fp(j);
    if (xj == x) return fj;
    const double slope = (fp(j + 1) - fj) / (xp(j + 1) - xj);
    return slope * (x - xj) + fj;
}

struct ap_args {
    const uint8_t* tp;          // (n, niou) 0/1
    const u64* keys;            // sorted
    const int* vals;            // detection index per sorted position
    const int* unique;          // classes with labels, ascending
    const int* nt;              // labels per such class
    long n;
    int niou;
    double eps;
    int* tpc;                   // workspace (niou, n): cumulative tp at each sorted position, per IoU threshold
    double* env;                // workspace (niou, n): precision envelope
    double* ap;                 // out (n_unique, niou)
    double* p_curve;            // out (n_unique, 1000)
    double* r_curve;            // out (n_unique, 1000)
};

// grid (n_unique, niou)
__global__ void __launch_bounds__(AP_THREADS) k_ap_curves(const ap_args a)
{
    __shared__ int s_i[AP_WAVES];
    __shared__ double s_d[AP_WAVES];
    __shared__ long s_seg[2];
    __shared__ double s_y[AP_X];
    const int ci = blockIdx.x, j = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cls = a.unique[ci];
    if (threadIdx.x == 0) {
        const long s = lower_bound_keys(a.keys, a.n, (u64)(uint32_t)cls << 32);
        s_seg[0] = s;
        s_seg[1] = lower_bound_keys(a.keys, a.n, ((u64)(uint32_t)cls + 1) << 32) - s;
    }
    __syncthreads();
    const long start = s_seg[0], L = s_seg[1];
    if (L == 0) {                                                          // metrics.py:51-52: the rows stay zero
        if (threadIdx.x == 0) a.ap[(size_t)ci * a.niou + j] = 0.0;
        if (j == 0)
            for (int q = threadIdx.x; q < AP_PX; q += AP_THREADS) {
                a.p_curve[(size_t)ci * AP_PX + q] = 0.0;
                a.r_curve[(size_t)ci * AP_PX + q] = 0.0;
            }
        return;
    }
    int* __restrict__ tpc = a.tpc + (size_t)j * a.n + start;
    double* __restrict__ env = a.env + (size_t)j * a.n + start;
    const long chunks = (L + AP_THREADS - 1) / AP_THREADS;

    // forward: cumulative true positives (metrics.py:56)
    int carry = 0;
    for (long ch = 0; ch < chunks; ++ch) {
        const long k = ch * AP_THREADS + threadIdx.x;
        int v = k < L ? (a.tp[(size_t)a.vals[start + k] * a.niou + j] != 0) : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        if (lane == 63) s_i[wave] = v;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < AP_WAVES; ++w) { const int c = s_i[w]; if (w < wave) before += c; all += c; }
        __syncthreads();
        if (k < L) tpc[k] = carry + before + v;
        carry += all;
    }
    __syncthreads();

    // backward: precision = tpc / (tpc + fpc) = tpc / (position + 1) (:63), envelope = running maximum from the end,
    // starting from the 0.0 sentinel (:100-103)
    double tail = 0.0;
    for (long ch = chunks - 1; ch >= 0; --ch) {
        const long k = ch * AP_THREADS + threadIdx.x;
        double v = k < L ? (double)tpc[k] / (double)(k + 1) : -INFINITY;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_down(v, o, 64); if (lane + o < 64) v = fmax(v, t); }
        if (lane == 0) s_d[wave] = v;
        __syncthreads();
        double after = -INFINITY, all = -INFINITY;
#pragma unroll
        for (int w = 0; w < AP_WAVES; ++w) { const double c = s_d[w]; if (w > wave) after = fmax(after, c); all = fmax(all, c); }
        __syncthreads();
        if (k < L) env[k] = fmax(fmax(v, after), tail);
        tail = fmax(tail, all);
    }
    __syncthreads();

    const double denom = (double)a.nt[ci] + a.eps;                         // :59
    // mrec = [0, recall..., 1], mpre = [max(1, envelope[0]), envelope..., 0]  (:99-103)
    const double m0 = fmax(1.0, env[0]);
    auto mrec = [&](long i) { return i == 0 ? 0.0 : (i == L + 1 ? 1.0 : (double)tpc[i - 1] / denom); };
    auto mpre = [&](long i) { return i == 0 ? m0 : (i == L + 1 ? 0.0 : env[i - 1]); };
    if (threadIdx.x < AP_X) {
        const int q = threadIdx.x;
        const double x = q == AP_X - 1 ? 1.0 : (double)q * (1.0 / (AP_X - 1));   // np.linspace(0, 1, 101)
        s_y[q] = interp_at(x, L + 2, mrec, mpre, mpre(0), mpre(L + 1));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // np.trapz: sum over (x[i+1]-x[i]) * (y[i+1]+y[i]) / 2 in np.sum's order for 100 contiguous doubles
        // (eight running sums over strides of eight, combined pairwise, then the remaining four in order)
        double r[8];
        auto term = [&](int i) {
            const double x1 = i + 1 == AP_X - 1 ? 1.0 : (double)(i + 1) * (1.0 / (AP_X - 1));
            const double x0 = (double)i * (1.0 / (AP_X - 1));
            return (x1 - x0) * (s_y[i + 1] + s_y[i]) / 2.0;
        };
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = term(k);
        for (int i = 8; i < 96; i += 8)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += term(i + k);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = 96; i < 100; ++i) res += term(i);
        a.ap[(size_t)ci * a.niou + j] = res;
    }
    if (j != 0) return;

    // recall and precision at 1000 confidences (:60, :64): np.interp(-px, -conf, curve, left)
    const u64* __restrict__ keys = a.keys + start;
    auto xp = [&](long k) { return -(double)conf_of_key(keys[k]); };
    auto rec = [&](long k) { return (double)tpc[k] / denom; };
    auto pre = [&](long k) { return (double)tpc[k] / (double)(k + 1); };
    const double step = 1.0 / (double)(AP_PX - 1);                         // np.linspace(0, 1, 1000)
    for (int q = threadIdx.x; q < AP_PX; q += AP_THREADS) {
        const double x = -(q == AP_PX - 1 ? 1.0 : (double)q * step);
        a.r_curve[(size_t)ci * AP_PX + q] = interp_at(x, L, xp, rec, 0.0, rec(L - 1));
        a.p_curve[(size_t)ci * AP_PX + q] = interp_at(x, L, xp, pre, 1.0, pre(L - 1));
    }
}

// F1 curve, the confidence index of the best mean F1, and the five per-class vectors at that index (metrics.py:73-86).
// One workgroup; out5 is (5, n_unique): tp, fp, p, r, f1.
__global__ void __launch_bounds__(AP_THREADS) k_ap_final(const double* __restrict__ p_curve, const double* __restrict__ r_curve,
                                                         const int* __restrict__ nt, int n_unique, double eps,
                                                         double* __restrict__ out5, int* __restrict__ best_index)
{
    __shared__ double s_v[AP_WAVES];
    __shared__ int s_q[AP_WAVES];
    __shared__ int s_best;
    const int q = threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mean = -INFINITY;
    if (q < AP_PX) {
        double sum = 0.0;
        for (int ci = 0; ci < n_unique; ++ci) {                            // np.mean(axis 0): rows added in order
            const double p = p_curve[(size_t)ci * AP_PX + q], r = r_curve[(size_t)ci * AP_PX + q];
            const double f1 = 2 * p * r / (p + r + eps);
            sum = ci == 0 ? f1 : sum + f1;
        }
        mean = sum / (double)n_unique;
    }
    // first index of the maximum
    double bv = mean;
    int bq = q < AP_PX ? q : 0x7FFFFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, 64);
        const int oq = __shfl_xor(bq, o, 64);
        if (ov > bv || (ov == bv && oq < bq)) { bv = ov; bq = oq; }
    }
    if (lane == 0) { s_v[wave] = bv; s_q[wave] = bq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < AP_WAVES; ++w)
            if (s_v[w] > bv || (s_v[w] == bv && s_q[w] < bq)) { bv = s_v[w]; bq = s_q[w]; }
        s_best = bq;
        *best_index = bq;
    }
    __syncthreads();
    const int i = s_best;
    for (int ci = threadIdx.x; ci < n_unique; ci += AP_THREADS) {
        const double p = p_curve[(size_t)ci * AP_PX + i], r = r_curve[(size_t)ci * AP_PX + i];
        const double f1 = 2 * p * r / (p + r + eps);
        const double tp = rint(r * (double)nt[ci]);                        // :84
        const double fp = rint(tp / (p + eps) - tp);                       // :85
        out5[0 * (size_t)n_unique + ci] = tp;
        out5[1 * (size_t)n_unique + ci] = fp;
        out5[2 * (size_t)n_unique + ci] = p;
        out5[3 * (size_t)n_unique + ci] = r;
        out5[4 * (size_t)n_unique + ci] = f1;
    }
}

// ---- radix sort -----------------------------------------------------------------------------------------------------------
// Least-significant-digit radix sort of (key, value) pairs, 8 bits per pass, stable.  A pass is three launches: every
// workgroup counts the digits of its tile of RS_TILE keys; one workgroup turns the (digit, workgroup) counts into
// exclusive offsets (digit-major: all of digit 0's tiles, then digit 1's, ...); every workgroup goes over its tile again,
// 256 keys at a time in input order, ranks each key among the equal digits before it (within the wave by ballots, across
// waves and earlier chunks through LDS counters) and writes it to its digit's offset plus its rank.
#define RS_THREADS 256
#define RS_CHUNKS 16
#define RS_TILE (RS_THREADS * RS_CHUNKS)

__global__ void __launch_bounds__(RS_THREADS) k_rs_count(const u64* __restrict__ keys, long n, int shift, uint32_t* __restrict__ hist,
                                                         uint32_t nwg)
{
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const long first = (long)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const long i = first + c * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nwg + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive prefix sum of `total` counters in place, by one workgroup
__global__ void __launch_bounds__(AP_THREADS) k_rs_offsets(uint32_t* __restrict__ hist, size_t total)
{
    __shared__ uint32_t s_wave[AP_WAVES];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (size_t base = 0; base < total; base += AP_THREADS) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < total ? hist[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < total) hist[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == AP_THREADS - 1) s_carry = before + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const u64* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                           u64* __restrict__ keys_out, int* __restrict__ vals_out, long n, int shift,
                                                           const uint32_t* __restrict__ hist, uint32_t nwg)
{
    __shared__ uint32_t s_base[256];                 // where this tile's keys of each digit start in the output
    __shared__ uint32_t s_cnt[RS_THREADS / 64][256]; // per chunk: keys of each digit per wave, then their offsets inside the digit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_base[threadIdx.x] = hist[(size_t)threadIdx.x * nwg + blockIdx.x];
    const long first = (long)blockIdx.x * RS_TILE;
    for (int c = 0; c < RS_CHUNKS; ++c) {
        if (first + c * RS_THREADS >= n) break;                             // workgroup-uniform
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; ++w) s_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const long i = first + c * RS_THREADS + threadIdx.x;
        const bool live = i < n;
        const u64 key = live ? keys_in[i] : 0;
        const int val = live ? vals_in[i] : 0;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        u64 peers = __ballot(live);                                         // the live lanes of this wave with the same digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = (d >> bit) & 1u;
            const u64 bm = __ballot(one);
            peers &= one ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (live && rank == 0) s_cnt[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        {       // thread t owns digit t: the waves' counts become offsets behind what earlier chunks and waves put there
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < RS_THREADS / 64; ++w) {
                const uint32_t cnt = s_cnt[w][threadIdx.x];
                s_cnt[w][threadIdx.x] = run;
                run += cnt;
            }
            const uint32_t start = s_base[threadIdx.x];
            s_base[threadIdx.x] = start + run;
#pragma unroll
            for (int w = 0; w < RS_THREADS / 64; ++w) s_cnt[w][threadIdx.x] += start;
        }
        __syncthreads();
        if (live) {
            const size_t pos = (size_t)s_cnt[wave][d] + rank;
            if (pos < (size_t)n) {                                          // always, while counts and keys agree
                keys_out[pos] = key;
                vals_out[pos] = val;
            }
        }
        __syncthreads();
    }
}

// Sorts n pairs by the low `bits` bits of their keys; a and b are two buffers of n pairs, the input is in a.  Returns which of
// the two holds the result (0: a, 1: b), or -1.
int radix_sort_pairs(u64* keys_a, int* vals_a, u64* keys_b, int* vals_b, long n, int bits, uint32_t* hist, hipStream_t st)
{
    const uint32_t nwg = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    int where = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        const u64* ki = where ? keys_b : keys_a;
        const int* vi = where ? vals_b : vals_a;
        u64* ko = where ? keys_a : keys_b;
        int* vo = where ? vals_a : vals_b;
        hipLaunchKernelGGL(k_rs_count, dim3(nwg), dim3(RS_THREADS), 0, st, ki, n, shift, hist, nwg);
        hipLaunchKernelGGL(k_rs_offsets, dim3(1), dim3(AP_THREADS), 0, st, hist, (size_t)256 * nwg);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nwg), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, shift, (const uint32_t*)hist, nwg);
        where ^= 1;
    }
    return dh_launch_status() == DH_OK ? where : -1;
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

struct ap_layout {
    size_t keys_in, keys_out, vals_in, vals_out, tpc, env, sort_tmp, sort_bytes, total;
};

bool ap_make_layout(int64_t n, int niou, ap_layout& l)
{
    const size_t sort_bytes = sizeof(uint32_t) * 256 * (size_t)((n + RS_TILE - 1) / RS_TILE);     // (digit, workgroup) counters
    size_t at = 0;
    l.keys_in = at;  at += align256(sizeof(u64) * n);
    l.keys_out = at; at += align256(sizeof(u64) * n);
    l.vals_in = at;  at += align256(sizeof(int) * n);
    l.vals_out = at; at += align256(sizeof(int) * n);
    l.tpc = at;      at += align256(sizeof(int) * n * niou);
    l.env = at;      at += align256(sizeof(double) * n * niou);
    l.sort_tmp = at; at += align256(sort_bytes);
    l.sort_bytes = sort_bytes;
    l.total = at + 256;
    return true;
}

}  // namespace

extern "C" int dh_ap_classes(const double* target_cls, int64_t m, int class_cap, int* unique, int* nt, int* meta, int* bins,
                             void* stream)
{
    if (m < 0 || class_cap < 1 || class_cap > (1 << 24) || !unique || !nt || !meta || !bins || (m > 0 && !target_cls))
        return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dh_fill_async(bins, 0, sizeof(int) * class_cap, st) != DH_OK) return DH_ERR_LAUNCH;
    if (dh_fill_async(meta, 0, sizeof(int) * 2, st) != DH_OK) return DH_ERR_LAUNCH;
    if (m > 0)
        hipLaunchKernelGGL(k_ap_hist, dim3((unsigned)((m + AP_THREADS * AP_HIST_PER_THREAD - 1) / (AP_THREADS * AP_HIST_PER_THREAD))),
                           dim3(AP_THREADS), 0, st, target_cls, (long)m, class_cap, bins, meta);
    hipLaunchKernelGGL(k_ap_unique, dim3(1), dim3(AP_THREADS), 0, st, bins, class_cap, unique, nt, meta);
    return dh_launch_status();
}

extern "C" size_t dh_ap_per_class_workspace(int64_t n, int niou)
{
    ap_layout l;
    if (n < 0 || niou < 1 || !ap_make_layout(n, niou, l)) return 0;
    return l.total;
}

extern "C" int dh_ap_per_class(const uint8_t* tp, const float* conf, const float* pred_cls, int64_t n, int niou,
                               const int* unique, const int* nt, int n_unique, int class_cap, double eps, double* ap,
                               double* p_curve, double* r_curve, double* out5, int* best_index, void* workspace,
                               size_t workspace_bytes, void* stream)
{
    if (n < 0 || n >= (1ll << 31) || niou < 1 || niou > 65535 || n_unique < 0 || class_cap < 1) return DH_ERR_ARG;
    if (n_unique == 0) return DH_OK;
    if (!unique || !nt || !ap || !p_curve || !r_curve || !out5 || !best_index || !workspace) return DH_ERR_ARG;
    if (n > 0 && (!tp || !conf || !pred_cls)) return DH_ERR_ARG;
    if (!dh_aligned16(workspace)) return DH_ERR_ALIGN;
    ap_layout l;
    if (!ap_make_layout(n, niou, l)) return DH_ERR_LAUNCH;
    if (workspace_bytes < l.total) return DH_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    u64* keys_in = (u64*)(ws + l.keys_in);
    u64* keys_out = (u64*)(ws + l.keys_out);
    int* vals_in = (int*)(ws + l.vals_in);
    int* vals_out = (int*)(ws + l.vals_out);
    if (n > 0) {
        hipLaunchKernelGGL(k_ap_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, conf, pred_cls, (long)n, class_cap,
                           keys_in, vals_in);
        int class_bits = 1;                                      // classes 0 .. class_cap (the last: no valid class)
        while ((1ll << class_bits) <= (long long)class_cap) ++class_bits;
        const int where = radix_sort_pairs(keys_in, vals_in, keys_out, vals_out, (long)n, 32 + class_bits, (uint32_t*)(ws + l.sort_tmp), st);
        if (where < 0) return DH_ERR_LAUNCH;
        if (where == 0) { keys_out = keys_in; vals_out = vals_in; }
    }
    ap_args a;
    a.tp = tp; a.keys = keys_out; a.vals = vals_out; a.unique = unique; a.nt = nt; a.n = (long)n; a.niou = niou; a.eps = eps;
    a.tpc = (int*)(ws + l.tpc); a.env = (double*)(ws + l.env); a.ap = ap; a.p_curve = p_curve; a.r_curve = r_curve;
    hipLaunchKernelGGL(k_ap_curves, dim3((unsigned)n_unique, (unsigned)niou), dim3(AP_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_ap_final, dim3(1), dim3(AP_THREADS), 0, st, (const double*)p_curve, (const double*)r_curve, nt, n_unique,
                       eps, out5, best_index);
    return dh_launch_status();
}
