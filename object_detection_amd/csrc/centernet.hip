// CenterNet head: heat-map peak filter, detection decode, and target-map drawing.
//
// Reference (object_detection/detectors/centernet/dataset.py):
//   :164-169 pool_nms      heat * (maxpool3x3(heat) == heat)
//   :172-216 decode_bbox   tanh of the filtered heat, best class per cell (first on ties), keep
//                          conf > confidence, box = (cell + offset -/+ size/2) / (W, H), rows in cell order
//   :96-161  target maps   boxes scaled to the stride-4 grid and clipped, centre cell, class heat = 1,
//                          size, offset = centre - cell, mask; later labels overwrite earlier ones
// Maps are (B, C, W, H) with H contiguous; the first spatial axis is x throughout, as in the reference.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

// value of pool_nms at one element: v where v equals the max of its KxK neighbourhood (-inf padding), else v*0
__device__ __forceinline__ float peak_value(const float* __restrict__ plane, int W, int H, int x, int y, int pad)
{
    const float v = plane[(size_t)x * H + y];
    float m = v;
    const int x0 = x - pad < 0 ? 0 : x - pad, x1 = x + pad >= W ? W - 1 : x + pad;
    const int y0 = y - pad < 0 ? 0 : y - pad, y1 = y + pad >= H ? H - 1 : y + pad;
    for (int xx = x0; xx <= x1; ++xx)
        for (int yy = y0; yy <= y1; ++yy) m = fmaxf(m, plane[(size_t)xx * H + yy]);
    return v * ((m == v) ? 1.0f : 0.0f);
}

__global__ void __launch_bounds__(256) k_cn_pool_nms(const float* __restrict__ heat, float* __restrict__ out, int planes,
                                                     int W, int H, int pad)
{
    const long cells = (long)W * H;
    const long total = cells * planes;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / cells;
        const int cell = (int)(i - pl * cells);
        out[i] = peak_value(heat + pl * cells, W, H, cell / H, cell % H, pad);
    }
}

// ---- 3x3 window (the only size the reference uses), one wave per (plane, x, segment of y) -----------------------------
// Lane j of a wave sits on y = y0 - 1 + j: lanes 1..62 produce output, lanes 0 and 63 are the halo.  Every lane loads
// its column's three values (rows x-1, x, x+1: three coalesced 256-byte wave loads), takes their maximum, and gets the
// two neighbouring columns' maxima by shuffle: 3 loads + 2 shuffles per element instead of 9 loads.
#define CN_SEG 62
#define CN4_ROWS 4          // rows per thread, 16-byte-vector peak filter
#define CN4_DEC_ROWS 2      // rows per thread, 16-byte-vector best-class pass (more workgroups per image)

struct cn_tile {
    int x, y;        // this lane's cell (y may be -1 or >= H on halo lanes)
    bool out;        // lane produces an output (interior lane, y inside the map)
};

__device__ __forceinline__ cn_tile cn_tile_of(int w, const dh_fastdiv segs, int H)
{
    const int lane = threadIdx.x & 63;
    const int x = (int)dh_div((uint32_t)w, segs);
    const int seg = w - x * (int)segs.d;
    cn_tile t;
    t.x = x;
    t.y = seg * CN_SEG - 1 + lane;
    t.out = lane >= 1 && lane <= CN_SEG && t.y < H;
    return t;
}

// pool_nms value of the lane's cell on `plane`: v where v is the maximum of its 3x3 window (-inf outside the map), else v*0
__device__ __forceinline__ float cn_peak3(const float* __restrict__ plane, int W, int H, const cn_tile t)
{
    const bool in = t.y >= 0 && t.y < H;
    const float* __restrict__ p = plane + (size_t)t.x * H + t.y;
    const float v = in ? p[0] : -INFINITY;
    const float up = (in && t.x > 0) ? p[-H] : -INFINITY;
    const float dn = (in && t.x < W - 1) ? p[H] : -INFINITY;
    const float col = fmaxf(v, fmaxf(up, dn));
    const float l = __shfl_up(col, 1, 64), r = __shfl_down(col, 1, 64);
    const float m = fmaxf(col, fmaxf(l, r));
    return v * ((m == v) ? 1.0f : 0.0f);
}

__global__ void __launch_bounds__(256) k_cn_pool_nms3(const float* __restrict__ heat, float* __restrict__ out, int planes,
                                                      int W, int H, const dh_fastdiv segs)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= W * (int)segs.d) return;                       // wave-uniform
    const cn_tile t = cn_tile_of(w, segs, H);
    const size_t cells = (size_t)W * H;
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        const float pv = cn_peak3(heat + pl * cells, W, H, t);
        if (t.out) out[pl * cells + (size_t)t.x * H + t.y] = pv;
    }
}

// ---- best class of a cell: conf = max_c tanh(pool_nms(heat)[c]), cls = the first class that reaches it ---------------
// tanh does not decrease, so the maximum is tanh of the largest filtered value m1, and the first class holding m1 is the
// answer unless a smaller filtered value rounds to the same tanh.  The class loop therefore only tracks m1, its first
// class, and m2 = the largest value below m1 (no tanh, no branches); afterwards tanhf(m2) == tanhf(m1) sends the cell
// to a scan over the classes (two heat values within a few ulps of each other, or a saturated tanh: rare).
__device__ __forceinline__ void cn_track(float pv, int c, float& m1, float& m2, int& c1)
{
    const bool gt = pv > m1;
    const float lower = (pv < m1) ? fmaxf(m2, pv) : m2;
    m2 = gt ? m1 : lower;
    c1 = gt ? c : c1;
    m1 = gt ? pv : m1;
}

__device__ __forceinline__ void cn_resolve(const float* __restrict__ img, int C, int W, int H, int x, int y, float m1,
                                           float m2, int c1, float& conf, int& cls)
{
    conf = tanhf(m1);
    cls = c1;
    if (tanhf(m2) == conf) {
        const size_t cells = (size_t)W * H;
        for (int c = 0; c < c1; ++c)
            if (tanhf(peak_value(img + (size_t)c * cells, W, H, x, y, 1)) == conf) { cls = c; break; }
    }
}

__global__ void __launch_bounds__(256) k_cn_best_class3(const float* __restrict__ heat, int C, int W, int H,
                                                        const dh_fastdiv segs, float* __restrict__ conf_out,
                                                        int* __restrict__ cls_out)
{
    const int b = blockIdx.y;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= W * (int)segs.d) return;                       // wave-uniform
    const cn_tile t = cn_tile_of(w, segs, H);
    const size_t cells = (size_t)W * H;
    const float* __restrict__ img = heat + (size_t)b * C * cells;
    float m1 = -INFINITY, m2 = -INFINITY;
    int c1 = 0;
    for (int c = 0; c < C; ++c) cn_track(cn_peak3(img + (size_t)c * cells, W, H, t), c, m1, m2, c1);
    if (t.out) {
        float conf;
        int cls;
        cn_resolve(img, C, W, H, t.x, t.y, m1, m2, c1, conf, cls);
        conf_out[(size_t)b * cells + (size_t)t.x * H + t.y] = conf;
        cls_out[(size_t)b * cells + (size_t)t.x * H + t.y] = cls;
    }
}

// ---- 3x3 window on maps whose H is a multiple of four: 16-byte vectors, R rows per thread ----------------------------
// A workgroup covers whole rows: lpr = H/4 threads per row, rpb = 256/lpr row groups of R consecutive rows (threads past
// rpb*lpr idle).  A thread loads its column vector of rows x0-1 .. x0+R (R+2 aligned 16-byte loads for R output rows),
// takes the vertical 3-maxima, and trades the outer components of each with its left and right neighbours through LDS
// (one barrier per plane).
template <int R>
__device__ __forceinline__ void cn4_load(const float4* __restrict__ plane, int W, int lpr, int x0, int y4, float4 (&v)[R + 2])
{
    // rows past the map's edge repeat the edge row (the 3-maximum is the same as with -inf padding), and idle threads
    // read valid rows too: no load sits behind a branch
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
        const int x = min(max(x0 - 1 + r, 0), W - 1);
        v[r] = plane[(size_t)x * lpr + y4];
    }
}

__device__ __forceinline__ float cn_keep(float v, float m) { return v * ((m == v) ? 1.0f : 0.0f); }

// s_w / s_x: [R][256] floats each.  Every thread of the workgroup must call this (barrier inside).
template <int R>
__device__ __forceinline__ void cn4_peaks(const float4 (&v)[R + 2], float* __restrict__ s_w, float* __restrict__ s_x,
                                          int y4, int lpr, float4 (&o)[R])
{
    const int t = threadIdx.x;
    float4 col[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        col[r].x = fmaxf(v[r + 1].x, fmaxf(v[r].x, v[r + 2].x));
        col[r].y = fmaxf(v[r + 1].y, fmaxf(v[r].y, v[r + 2].y));
        col[r].z = fmaxf(v[r + 1].z, fmaxf(v[r].z, v[r + 2].z));
        col[r].w = fmaxf(v[r + 1].w, fmaxf(v[r].w, v[r + 2].w));
        s_w[r * 256 + t] = col[r].w;
        s_x[r * 256 + t] = col[r].x;
    }
    __syncthreads();
    const int tl = t > 0 ? t - 1 : 0, tr = t < 255 ? t + 1 : 255;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float l = y4 > 0 ? s_w[r * 256 + tl] : -INFINITY;
        const float rr = y4 < lpr - 1 ? s_x[r * 256 + tr] : -INFINITY;
        const float4 c = col[r], u = v[r + 1];
        o[r].x = cn_keep(u.x, fmaxf(fmaxf(c.x, c.y), l));
        o[r].y = cn_keep(u.y, fmaxf(fmaxf(c.x, c.y), c.z));
        o[r].z = cn_keep(u.z, fmaxf(fmaxf(c.y, c.z), c.w));
        o[r].w = cn_keep(u.w, fmaxf(fmaxf(c.z, c.w), rr));
    }
}

struct cn4_pos {
    int x0, y4;
    bool active;
};

template <int R>
__device__ __forceinline__ cn4_pos cn4_pos_of(const dh_fastdiv lpr, int rpb, int W)
{
    const int g = (int)dh_div(threadIdx.x, lpr);
    cn4_pos p;
    p.y4 = (int)threadIdx.x - g * (int)lpr.d;
    p.x0 = (blockIdx.x * rpb + g) * R;
    p.active = g < rpb && p.x0 < W;
    return p;
}

template <int R>
__global__ void __launch_bounds__(256) k_cn_pool_nms3v(const float* __restrict__ heat, float* __restrict__ out, int planes,
                                                       int W, int H, const dh_fastdiv lpr, int rpb)
{
    __shared__ float s_w[R * 256], s_x[R * 256];
    const cn4_pos p = cn4_pos_of<R>(lpr, rpb, W);
    const size_t cells = (size_t)W * H;
    for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        float4 v[R + 2], o[R];
        cn4_load<R>((const float4*)(heat + pl * cells), W, (int)lpr.d, p.x0, p.y4, v);
        cn4_peaks<R>(v, s_w, s_x, p.y4, (int)lpr.d, o);
        dh_f4* __restrict__ q = reinterpret_cast<dh_f4*>(out + pl * cells);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (p.active && p.x0 + r < W) {
                const dh_f4 ov = {o[r].x, o[r].y, o[r].z, o[r].w};
                __builtin_nontemporal_store(ov, &q[(size_t)(p.x0 + r) * lpr.d + p.y4]);
            }
        __syncthreads();                                    // the next plane rewrites the exchange arrays
    }
}

template <int R>
__global__ void __launch_bounds__(256) k_cn_best_class3v(const float* __restrict__ heat, int C, int W, int H,
                                                         const dh_fastdiv lpr, int rpb, float* __restrict__ conf_out,
                                                         int* __restrict__ cls_out)
{
    __shared__ float s_w[2][R * 256], s_x[2][R * 256];      // two sets: one barrier per class
    const int b = blockIdx.y;
    const cn4_pos p = cn4_pos_of<R>(lpr, rpb, W);
    const size_t cells = (size_t)W * H;
    const float* __restrict__ img = heat + (size_t)b * C * cells;
    float4 m1[R], m2[R];
    int4 c1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m1[r] = m2[r] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        c1[r] = make_int4(0, 0, 0, 0);
    }
    float4 v[R + 2];
    cn4_load<R>((const float4*)img, W, (int)lpr.d, p.x0, p.y4, v);
    for (int c = 0; c < C; ++c) {
        float4 nv[R + 2], o[R];
        const int cn = c + 1 < C ? c + 1 : c;                                           // the next class map, loaded ahead
        cn4_load<R>((const float4*)(img + (size_t)cn * cells), W, (int)lpr.d, p.x0, p.y4, nv);
        cn4_peaks<R>(v, s_w[c & 1], s_x[c & 1], p.y4, (int)lpr.d, o);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cn_track(o[r].x, c, m1[r].x, m2[r].x, c1[r].x);
            cn_track(o[r].y, c, m1[r].y, m2[r].y, c1[r].y);
            cn_track(o[r].z, c, m1[r].z, m2[r].z, c1[r].z);
            cn_track(o[r].w, c, m1[r].w, m2[r].w, c1[r].w);
        }
#pragma unroll
        for (int r = 0; r < R + 2; ++r) v[r] = nv[r];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (p.active && p.x0 + r < W) {
            const int x = p.x0 + r, y = p.y4 * 4;
            float4 conf;
            int4 cls;
            cn_resolve(img, C, W, H, x, y + 0, m1[r].x, m2[r].x, c1[r].x, conf.x, cls.x);
            cn_resolve(img, C, W, H, x, y + 1, m1[r].y, m2[r].y, c1[r].y, conf.y, cls.y);
            cn_resolve(img, C, W, H, x, y + 2, m1[r].z, m2[r].z, c1[r].z, conf.z, cls.z);
            cn_resolve(img, C, W, H, x, y + 3, m1[r].w, m2[r].w, c1[r].w, conf.w, cls.w);
            const size_t at = (size_t)b * cells / 4 + (size_t)x * lpr.d + p.y4;
            ((float4*)conf_out)[at] = conf;
            ((int4*)cls_out)[at] = cls;
        }
}

// order-preserving compaction of the cells above the confidence threshold: workgroups of CN_CHUNK cells count their
// hits, then every workgroup adds up the counts of the chunks before it -- of its own image and of all images before it,
// so that the rows of the whole batch end up packed, image after image -- and writes its own rows
#define CN_CHUNK 1024

__global__ void __launch_bounds__(CN_CHUNK) k_cn_count(const float* __restrict__ conf, int cells, float confidence,
                                                       int* __restrict__ chunk_counts)
{
    __shared__ int s_w[CN_CHUNK / 64];
    const int b = blockIdx.y, cell = blockIdx.x * CN_CHUNK + threadIdx.x;
    const bool take = cell < cells && conf[(size_t)b * cells + cell] > confidence;
    const unsigned long long bm = __ballot(take);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int all = 0;
#pragma unroll
        for (int w = 0; w < CN_CHUNK / 64; ++w) all += s_w[w];
        chunk_counts[(size_t)b * gridDim.x + blockIdx.x] = all;
    }
}

__global__ void __launch_bounds__(CN_CHUNK) k_cn_emit(const float* __restrict__ conf, const int* __restrict__ cls,
                                                      const float* __restrict__ size, const float* __restrict__ offset,
                                                      int W, int H, float confidence, const int* __restrict__ chunk_counts,
                                                      float* __restrict__ det, int* __restrict__ counts)
{
    __shared__ int s_w[CN_CHUNK / 64], s_prev[CN_CHUNK / 64], s_prev_img[CN_CHUNK / 64];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cells = W * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell = chunk * CN_CHUNK + threadIdx.x;
    int prev = 0, prev_img = 0;                      // hits in the chunks before this one: of the batch / of this image
    const int own_first = b * (int)gridDim.x;
    for (int k = threadIdx.x; k < own_first + chunk; k += CN_CHUNK) {
        const int c = chunk_counts[k];
        prev += c;
        prev_img += k >= own_first ? c : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        prev += __shfl_xor(prev, o, 64);
        prev_img += __shfl_xor(prev_img, o, 64);
    }
    float cf = 0.0f;
    bool take = false;
    if (cell < cells) {
        cf = conf[(size_t)b * cells + cell];
        take = cf > confidence;                                                       // dataset.py:189
    }
    const unsigned long long bm = __ballot(take);
    if (lane == 0) { s_w[wave] = __popcll(bm); s_prev[wave] = prev; s_prev_img[wave] = prev_img; }
    __syncthreads();
    int before = 0, all = 0;                          // rows of the batch before this wave's; hits of this image so far
#pragma unroll
    for (int w = 0; w < CN_CHUNK / 64; ++w) {
        const int c = s_w[w];
        if (w < wave) before += c;
        all += c + s_prev_img[w];
        before += s_prev[w];
    }
    if (take) {
        const int pos = before + __popcll(bm & ((1ull << lane) - 1ull));
        const int x = cell / H, y = cell - x * H;
        const float* __restrict__ sz = size + (size_t)b * 2 * cells;
        const float* __restrict__ of = offset + (size_t)b * 2 * cells;
        const float cx = (float)x + of[cell], cy = (float)y + of[cells + cell];        // dataset.py:197-198
        const float hw = sz[cell] / 2, hh = sz[cells + cell] / 2;
        const float fw = (float)W, fh = (float)H;
        float* __restrict__ d = det + (size_t)pos * 6;
        d[0] = __fdiv_rn(cx - hw, fw);                                                // dataset.py:200-205
        d[1] = __fdiv_rn(cy - hh, fh);
        d[2] = __fdiv_rn(cx + hw, fw);
        d[3] = __fdiv_rn(cy + hh, fh);
        d[4] = cf;
        d[5] = (float)cls[(size_t)b * cells + cell];
    }
    if (chunk == (int)gridDim.x - 1 && threadIdx.x == 0) counts[b] = all;
}

// ---- target maps.  labels: (B, Lmax, 5) [x1,y1,x2,y2,cls] input pixels, nlabels (B).
struct cn_label_cell {
    int x, y, cls;
    float w, h, ox, oy;
    bool ok;
};

__device__ __forceinline__ cn_label_cell label_cell(const float* __restrict__ l, float in_w, float in_h, int W, int H, int C)
{
    cn_label_cell r;
    const float x1 = fminf(fmaxf(__fdiv_rn(l[0], in_w) * (float)W, 0.0f), (float)(W - 1));   // dataset.py:119-123
    const float x2 = fminf(fmaxf(__fdiv_rn(l[2], in_w) * (float)W, 0.0f), (float)(W - 1));
    const float y1 = fminf(fmaxf(__fdiv_rn(l[1], in_h) * (float)H, 0.0f), (float)(H - 1));   // dataset.py:124-128
    const float y2 = fminf(fmaxf(__fdiv_rn(l[3], in_h) * (float)H, 0.0f), (float)(H - 1));
    r.w = x2 - x1;
    r.h = y2 - y1;
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
    r.x = (int)cx;
    r.y = (int)cy;
    r.ox = cx - (float)r.x;
    r.oy = cy - (float)r.y;
    r.cls = (int)l[4];
    r.ok = r.w > 0.0f && r.h > 0.0f && r.cls >= 0 && r.cls < C;                               // dataset.py:135
    return r;
}

__global__ void __launch_bounds__(256) k_cn_targets_mark(const float* __restrict__ labels, const int* __restrict__ nlabels,
                                                         int Lmax, float in_w, float in_h, int C, int W, int H,
                                                         float* __restrict__ t_cls, float* __restrict__ t_mask,
                                                         int* __restrict__ owner)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlabels[b] || i >= Lmax) return;
    const cn_label_cell r = label_cell(labels + ((size_t)b * Lmax + i) * 5, in_w, in_h, W, H, C);
    if (!r.ok) return;
    const size_t cells = (size_t)W * H, cell = (size_t)r.x * H + r.y;
    t_cls[((size_t)b * C + r.cls) * cells + cell] = 1.0f;                 // dataset.py:57-60 (the only live effect)
    t_mask[(size_t)b * cells + cell] = 1.0f;
    atomicMax(owner + (size_t)b * cells + cell, i);                       // the last label on a cell owns size/offset
}

__global__ void __launch_bounds__(256) k_cn_targets_fill(const float* __restrict__ labels, const int* __restrict__ nlabels,
                                                         int Lmax, float in_w, float in_h, int C, int W, int H,
                                                         const int* __restrict__ owner, float* __restrict__ t_size,
                                                         float* __restrict__ t_off)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlabels[b] || i >= Lmax) return;
    const cn_label_cell r = label_cell(labels + ((size_t)b * Lmax + i) * 5, in_w, in_h, W, H, C);
    if (!r.ok) return;
    const size_t cells = (size_t)W * H, cell = (size_t)r.x * H + r.y;
    if (owner[(size_t)b * cells + cell] != i) return;
    t_size[((size_t)b * 2 + 0) * cells + cell] = r.w;                     // dataset.py:145
    t_size[((size_t)b * 2 + 1) * cells + cell] = r.h;
    t_off[((size_t)b * 2 + 0) * cells + cell] = r.ox;                     // dataset.py:146
    t_off[((size_t)b * 2 + 1) * cells + cell] = r.oy;
}

// the 16-byte-vector kernels need H % 4 == 0, a row within one workgroup, and an aligned base
bool cn4_ok(const void* p, int H) { return (H & 3) == 0 && H <= 1024 && ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int dh_centernet_pool_nms(const float* heat, float* out, int64_t planes, int W, int H, int kernel, void* stream)
{
    if (planes < 0 || W < 1 || H < 1 || kernel < 1 || (kernel & 1) == 0 || planes > (1 << 30)) return DH_ERR_ARG;
    if (planes == 0) return DH_OK;
    if (!heat || !out) return DH_ERR_ARG;
    if (kernel == 3 && cn4_ok(heat, H) && cn4_ok(out, H)) {
        const int lpr = H / 4, rpb = 256 / lpr;
        const dim3 grid((unsigned)(((W + CN4_ROWS - 1) / CN4_ROWS + rpb - 1) / rpb), (unsigned)(planes < 65535 ? planes : 65535));
        hipLaunchKernelGGL(k_cn_pool_nms3v<CN4_ROWS>, grid, dim3(256), 0, (hipStream_t)stream, heat, out, (int)planes, W, H,
                           dh_make_fastdiv((uint32_t)lpr), rpb);
        return dh_launch_status();
    }
    if (kernel == 3 && (long)W * ((H + CN_SEG - 1) / CN_SEG) < (1l << 30)) {
        const int segs = (H + CN_SEG - 1) / CN_SEG;
        const dim3 grid((unsigned)((W * segs + 3) / 4), (unsigned)(planes < 65535 ? planes : 65535));
        hipLaunchKernelGGL(k_cn_pool_nms3, grid, dim3(256), 0, (hipStream_t)stream, heat, out, (int)planes, W, H,
                           dh_make_fastdiv((uint32_t)segs));
        return dh_launch_status();
    }
    const long total = (long)planes * W * H;
    long g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(k_cn_pool_nms, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, heat, out, (int)planes, W, H,
                       (kernel - 1) / 2);
    return dh_launch_status();
}

extern "C" size_t dh_centernet_decode_workspace(int B, int W, int H)
{
    if (B <= 0 || W <= 0 || H <= 0) return 0;
    const size_t cells = (size_t)W * H, chunks = (cells + CN_CHUNK - 1) / CN_CHUNK;
    return (size_t)B * cells * (sizeof(float) + sizeof(int)) + (size_t)B * chunks * sizeof(int) + 768;
}

extern "C" int dh_centernet_decode(const float* heat, const float* size, const float* offset, int B, int C, int W, int H,
                                   double confidence, float* det, int* counts, void* workspace, size_t workspace_bytes,
                                   void* stream)
{
    if (B < 0 || B > 65535 || C < 1 || W < 1 || H < 1 || (long)W * H >= (1l << 30)) return DH_ERR_ARG;
    if (B == 0) return DH_OK;
    if (!heat || !size || !offset || !det || !counts || !workspace) return DH_ERR_ARG;
    if (workspace_bytes < dh_centernet_decode_workspace(B, W, H)) return DH_ERR_WORKSPACE;
    const size_t cells = (size_t)W * H;
    float* conf = (float*)workspace;
    int* cls = (int*)((char*)workspace + ((sizeof(float) * B * cells + 255) / 256) * 256);
    hipStream_t st = (hipStream_t)stream;
    if (cn4_ok(heat, H)) {
        const int lpr = H / 4, rpb = 256 / lpr;
        const dim3 grid((unsigned)(((W + CN4_DEC_ROWS - 1) / CN4_DEC_ROWS + rpb - 1) / rpb), B);
        hipLaunchKernelGGL(k_cn_best_class3v<CN4_DEC_ROWS>, grid, dim3(256), 0, st, heat, C, W, H,
                           dh_make_fastdiv((uint32_t)lpr), rpb, conf, cls);
    } else {
        const int segs = (H + CN_SEG - 1) / CN_SEG;
        hipLaunchKernelGGL(k_cn_best_class3, dim3((unsigned)((W * segs + 3) / 4), B), dim3(256), 0, st, heat, C, W, H,
                           dh_make_fastdiv((uint32_t)segs), conf, cls);
    }
    const unsigned chunks = (unsigned)((cells + CN_CHUNK - 1) / CN_CHUNK);
    int* chunk_counts = (int*)((char*)cls + ((sizeof(int) * B * cells + 255) / 256) * 256);
    hipLaunchKernelGGL(k_cn_count, dim3(chunks, B), dim3(CN_CHUNK), 0, st, conf, (int)cells, (float)confidence, chunk_counts);
    hipLaunchKernelGGL(k_cn_emit, dim3(chunks, B), dim3(CN_CHUNK), 0, st, conf, cls, size, offset, W, H, (float)confidence,
                       chunk_counts, det, counts);
    return dh_launch_status();
}

extern "C" int dh_centernet_targets(const float* labels, const int* nlabels, int B, int Lmax, double input_w, double input_h,
                                    int C, int W, int H, float* t_cls, float* t_size, float* t_offset, float* t_mask,
                                    int* owner_ws, void* stream)
{
    if (B < 0 || B > 65535 || Lmax < 0 || C < 1 || W < 1 || H < 1) return DH_ERR_ARG;
    if (B == 0) return DH_OK;
    if (!t_cls || !t_size || !t_offset || !t_mask || !owner_ws || !nlabels) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)W * H;
    if (dh_fill_async(t_cls, 0, sizeof(float) * B * C * cells, st) != DH_OK) return DH_ERR_LAUNCH;
    if (dh_fill_async(t_size, 0, sizeof(float) * B * 2 * cells, st) != DH_OK) return DH_ERR_LAUNCH;
    if (dh_fill_async(t_offset, 0, sizeof(float) * B * 2 * cells, st) != DH_OK) return DH_ERR_LAUNCH;
    if (dh_fill_async(t_mask, 0, sizeof(float) * B * cells, st) != DH_OK) return DH_ERR_LAUNCH;
    if (dh_fill_async(owner_ws, 0xFF, sizeof(int) * B * cells, st) != DH_OK) return DH_ERR_LAUNCH;
    if (Lmax == 0) return DH_OK;
    if (!labels) return DH_ERR_ARG;
    const dim3 grid((Lmax + 255) / 256, B);
    hipLaunchKernelGGL(k_cn_targets_mark, grid, dim3(256), 0, st, labels, nlabels, Lmax, (float)input_w, (float)input_h, C, W, H,
                       t_cls, t_mask, owner_ws);
    hipLaunchKernelGGL(k_cn_targets_fill, grid, dim3(256), 0, st, labels, nlabels, Lmax, (float)input_w, (float)input_h, C, W, H,
                       owner_ws, t_size, t_offset);
    return dh_launch_status();
}
