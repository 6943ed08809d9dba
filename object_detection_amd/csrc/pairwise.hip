// N x M pairwise IoU matrices.
//
// Reference: object_detection/yolov5/utils/metrics.py:242-264 (box_iou: xyxy, no eps)
//            object_detection/yolov5/utils/metrics.py:291-296 (wh_iou: width/height only)
// The kernel is write-bound (4*N*M bytes out against 16*(N+M) in).  A workgroup stages the row boxes of its tile and
// their areas in LDS once; a lane keeps FOUR consecutive column boxes (loaded as 16-byte vectors) and their areas in
// registers and computes eight rows of them, so a row of the tile leaves as one non-temporal 16-byte store per lane, 1 KB
// contiguous per wave.  Two tile shapes (measured with tools/micro/pairwise_bw.hip on 8192 x 8192 and 30000 x 3000):
//   strip  8 rows x 1024 columns: the four waves sit side by side and a row piece is 4 KB contiguous.  Taken when the
//          rows start on 128-byte lines (M % 32 == 0) and are long enough: 6.3 TB/s against 5.7 TB/s for the other shape.
//   tile   ROWS rows x 256 columns, the four waves take every fourth row.  32 rows when rows start on 128-byte lines;
//          8 rows otherwise: neighbouring tiles then share the lines at their edges, and short tiles, whose neighbours
//          run at the same time, lose least to that (4.8 TB/s against 3.6 TB/s at M = 3000).
// Tiles are numbered column-first in a one-dimensional grid (no 65 535 limit on either side).  When M is not a multiple
// of four, or a pointer is not 16-byte aligned, one column per lane and 4-byte accesses.  The quotient is the IEEE
// division the reference performs, bit for bit: either the division itself or, for tiles whose boxes are all "plain", the
// same refinement sequence without its range handling and on two pairs per instruction (pw_div2_plain; 47 -> 44 us at
// 8192 x 8192, 6.0 TB/s, against 42 us for the same stores with no arithmetic at all).
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

template <bool WH_ONLY> __device__ __forceinline__ float pw_area(float x, float y, float z, float w)
{
    return WH_ONLY ? x * y : (z - x) * (w - y);
}

// IoU of a staged row box r (area ra) and a column box c (area ca)
template <bool WH_ONLY>
__device__ __forceinline__ float pw_iou(const float4 r, float ra, float cx, float cy, float cz, float cw, float ca)
{
    float inter;
    if (WH_ONLY) inter = fminf(r.x, cx) * fminf(r.y, cy);
    else inter = fmaxf(fminf(r.z, cz) - fmaxf(r.x, cx), 0.0f) * fmaxf(fminf(r.w, cw) - fmaxf(r.y, cy), 0.0f);
    return __fdiv_rn(inter, ra + ca - inter);
}

typedef float dh_f2 __attribute__((ext_vector_type(2)));

// n / d of two pairs at once, correctly rounded for operands in the "plain" range (below).  The instruction sequence is the
// one the compiler expands an IEEE fp32 division into -- reciprocal estimate, one refinement of it, two refinements of the
// quotient -- without the three instructions around it that only matter outside that range (v_div_scale of numerator and
// denominator, v_div_fixup), and on two pairs per instruction (v_pk_fma_f32): 4.5 vector instructions per quotient instead of
// 10.  The kernel is bound by exactly those instructions at write-bound sizes (counters: 24 per element, the vector ALU busy
// for 41 of 47 us at 8192 x 8192).
__device__ __forceinline__ dh_f2 pw_div2_plain(const dh_f2 n, const dh_f2 d)
{
    const dh_f2 one = {1.0f, 1.0f};
    dh_f2 y = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const dh_f2 e = __builtin_elementwise_fma(-d, y, one);
    y = __builtin_elementwise_fma(e, y, y);
    dh_f2 q = n * y;
    dh_f2 r = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(r, y, q);
    r = __builtin_elementwise_fma(-d, q, n);
    return __builtin_elementwise_fma(r, y, q);
}

// A box is "plain" when every quotient it can take part in stays clear of the cases the scaling and fix-up instructions of a
// division exist for: finite, x2 > x1 and y2 > y1 (positive area, so no union is zero), every coordinate zero or between 2^-12
// and 2^16 in magnitude.  Then a non-zero intersection is at least 2^-70, a union between 2^-70 and 2^35, the quotient at
// least 2^-105: every intermediate of the sequence above is a normal number and it returns the bits of the IEEE division.
// Pixel and normalised coordinates are of that kind; a tile that holds any other box takes the IEEE division.
template <bool WH_ONLY>
__device__ __forceinline__ bool pw_plain(float x, float y, float z, float w)
{
    const float lo = 0x1p-12f, hi = 0x1p16f;
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z), aw = fabsf(w);
    // bitwise combination of the comparisons: straight-line code, no branch per term
    int ok = ((ax == 0.0f) | ((ax >= lo) & (ax <= hi))) & ((ay == 0.0f) | ((ay >= lo) & (ay <= hi)));
    if (WH_ONLY) return (ok & (x > 0.0f) & (y > 0.0f)) != 0;
    ok &= ((az == 0.0f) | ((az >= lo) & (az <= hi))) & ((aw == 0.0f) | ((aw >= lo) & (aw <= hi)));
    return (ok & (z > x) & (w > y)) != 0;      // (a NaN fails every comparison)
}

// VEC: four columns per lane (M % 4 == 0, 16-byte aligned pointers); else one column per lane, 64 columns per wave
// STRIP: the workgroup's tile is PW_ROWS_T rows x 1024 columns (else x 256 columns)
// PLAIN: tiles whose boxes are all plain take the short division (see pw_div2_plain).  Off for rows that do not start on
// 128-byte lines: there the kernel is held back by the lines neighbouring tiles share, not by arithmetic, and measured
// slower with the shorter sequence (101 against 78 us at 30000 x 3000; handing neighbouring tiles to one XCD changed nothing).
template <bool WH_ONLY, bool VEC, int PW_ROWS_T, bool STRIP, bool PLAIN>
__global__ void __launch_bounds__(256) k_pairwise_iou(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int N, int M, const dh_fastdiv col_tiles)
{
    constexpr int PW_ROWS = PW_ROWS_T;
    constexpr int PW_COLS = STRIP ? 1024 : 256;
    __shared__ float4 rows[PW_ROWS];
    __shared__ float rarea[PW_ROWS];
    const uint32_t rt = dh_div(blockIdx.x, col_tiles);
    const uint32_t ct = blockIdx.x - rt * col_tiles.d;
    const long row0 = (long)rt * PW_ROWS;
    bool plain = true;
    if (threadIdx.x < PW_ROWS) {
        const long r = row0 + threadIdx.x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) {
            if (WH_ONLY && VEC) { const float2 t = *reinterpret_cast<const float2*>(a + 2 * (size_t)r); v.x = t.x; v.y = t.y; }
            else if (WH_ONLY) { v.x = a[2 * (size_t)r]; v.y = a[2 * (size_t)r + 1]; }
            else if (VEC) v = *reinterpret_cast<const float4*>(a + 4 * (size_t)r);
            else { const float* p = a + 4 * (size_t)r; v = make_float4(p[0], p[1], p[2], p[3]); }
        }
        rows[threadIdx.x] = v;
        rarea[threadIdx.x] = pw_area<WH_ONLY>(v.x, v.y, v.z, v.w);
        plain = (r >= N) | pw_plain<WH_ONLY>(v.x, v.y, v.z, v.w);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = (N - row0) < PW_ROWS ? (int)(N - row0) : PW_ROWS;
    if (VEC) {
        const long col = (long)ct * PW_COLS + (STRIP ? threadIdx.x : lane) * 4;
        float cx[4], cy[4], cz[4], cw[4], ca[4];
        if (col >= M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { cx[j] = cy[j] = 1.0f; cz[j] = cw[j] = 2.0f; }
        } else if (WH_ONLY) {
            const float4 p0 = *reinterpret_cast<const float4*>(b + 2 * (size_t)col);
            const float4 p1 = *reinterpret_cast<const float4*>(b + 2 * (size_t)col + 4);
            cx[0] = p0.x; cy[0] = p0.y; cx[1] = p0.z; cy[1] = p0.w; cx[2] = p1.x; cy[2] = p1.y; cx[3] = p1.z; cy[3] = p1.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cz[j] = 0.f; cw[j] = 0.f; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 p = *reinterpret_cast<const float4*>(b + 4 * ((size_t)col + j));
                cx[j] = p.x; cy[j] = p.y; cz[j] = p.z; cw[j] = p.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ca[j] = pw_area<WH_ONLY>(cx[j], cy[j], cz[j], cw[j]);
            plain = plain & pw_plain<WH_ONLY>(cx[j], cy[j], cz[j], cw[j]);
        }
        // the barrier that publishes the staged rows also tells every thread whether all boxes of the tile are plain
        const bool all_plain = __syncthreads_and((PLAIN && plain) ? 1 : 0) != 0;
        if (col >= M) return;
        float* __restrict__ o = out + (size_t)row0 * M + col;
        if (PLAIN && all_plain) {
#pragma unroll 2
            for (int k = STRIP ? 0 : wave; k < nrows; k += STRIP ? 1 : 4) {
                const float4 r = rows[k];
                const float ra = rarea[k];
                float in[4], un[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (WH_ONLY) in[j] = fminf(r.x, cx[j]) * fminf(r.y, cy[j]);
                    else in[j] = fmaxf(fminf(r.z, cz[j]) - fmaxf(r.x, cx[j]), 0.0f) * fmaxf(fminf(r.w, cw[j]) - fmaxf(r.y, cy[j]), 0.0f);
                    un[j] = ra + ca[j] - in[j];
                }
                const dh_f2 n0 = {in[0], in[1]}, d0 = {un[0], un[1]}, n1 = {in[2], in[3]}, d1 = {un[2], un[3]};
                const dh_f2 q0 = pw_div2_plain(n0, d0), q1 = pw_div2_plain(n1, d1);
                const dh_f4 v = {q0.x, q0.y, q1.x, q1.y};
                __builtin_nontemporal_store(v, reinterpret_cast<dh_f4*>(o + (size_t)k * M));
            }
            return;
        }
#pragma unroll 2
        for (int k = STRIP ? 0 : wave; k < nrows; k += STRIP ? 1 : 4) {
            const float4 r = rows[k];
            const float ra = rarea[k];
            dh_f4 v;
            v.x = pw_iou<WH_ONLY>(r, ra, cx[0], cy[0], cz[0], cw[0], ca[0]);
            v.y = pw_iou<WH_ONLY>(r, ra, cx[1], cy[1], cz[1], cw[1], ca[1]);
            v.z = pw_iou<WH_ONLY>(r, ra, cx[2], cy[2], cz[2], cw[2], ca[2]);
            v.w = pw_iou<WH_ONLY>(r, ra, cx[3], cy[3], cz[3], cw[3], ca[3]);
            __builtin_nontemporal_store(v, reinterpret_cast<dh_f4*>(o + (size_t)k * M));
        }
    } else {
        __syncthreads();
        // the tile is 256 columns here too: every wave takes all rows of its own 64 columns
        const long col = (long)ct * PW_COLS + threadIdx.x;
        if (col >= M) return;
        float cx, cy, cz = 0.f, cw = 0.f;
        if (WH_ONLY) { cx = b[2 * (size_t)col]; cy = b[2 * (size_t)col + 1]; }
        else { const float* p = b + 4 * (size_t)col; cx = p[0]; cy = p[1]; cz = p[2]; cw = p[3]; }
        const float ca = pw_area<WH_ONLY>(cx, cy, cz, cw);
        float* __restrict__ o = out + (size_t)row0 * M + col;
#pragma unroll 4
        for (int k = 0; k < nrows; ++k)
            o[(size_t)k * M] = pw_iou<WH_ONLY>(rows[k], rarea[k], cx, cy, cz, cw, ca);
    }
}

int launch(bool wh, const float* a, const float* b, float* out, int64_t N, int64_t M, void* stream)
{
    if (N < 0 || M < 0 || N >= (1ll << 31) || M >= (1ll << 31)) return DH_ERR_ARG;
    if (N == 0 || M == 0) return DH_OK;
    if (!a || !b || !out) return DH_ERR_ARG;
    const bool vec = M % 4 == 0 && dh_aligned16(a) && dh_aligned16(b) && dh_aligned16(out);
    const bool lines = vec && M % 32 == 0 && (reinterpret_cast<uintptr_t>(out) & 127u) == 0;     // every row starts on a 128-byte line
    const int shape = lines && M >= 2048 ? 2 : (vec && !lines ? 1 : 0);                        // 2: strip, 1: 8-row tile, 0: 32-row tile
    const int rows = shape == 0 ? 32 : 8, cols = shape == 2 ? 1024 : 256;
    const int64_t row_tiles = (N + rows - 1) / rows, col_tiles = (M + cols - 1) / cols;
    if (row_tiles * col_tiles >= (1ll << 31)) return DH_ERR_ARG;      // beyond 4e12 pairs
    const dim3 grid((unsigned)(row_tiles * col_tiles));
    const dh_fastdiv ctd = dh_make_fastdiv((uint32_t)col_tiles);
    hipStream_t st = (hipStream_t)stream;
#define PW_LAUNCH(WH, VEC, ROWS, STRIP, PLAIN) \
    hipLaunchKernelGGL((k_pairwise_iou<WH, VEC, ROWS, STRIP, PLAIN>), grid, dim3(256), 0, st, a, b, out, (int)N, (int)M, ctd)
    if (wh) {
        if (shape == 2) PW_LAUNCH(true, true, 8, true, true);
        else if (shape == 1) PW_LAUNCH(true, true, 8, false, false);
        else if (vec) PW_LAUNCH(true, true, 32, false, true);
        else PW_LAUNCH(true, false, 32, false, false);
    } else {
        if (shape == 2) PW_LAUNCH(false, true, 8, true, true);
        else if (shape == 1) PW_LAUNCH(false, true, 8, false, false);
        else if (vec) PW_LAUNCH(false, true, 32, false, true);
        else PW_LAUNCH(false, false, 32, false, false);
    }
#undef PW_LAUNCH
    return dh_launch_status();
}

}  // namespace

extern "C" int dh_box_iou(const float* box1, const float* box2, float* out, int64_t N, int64_t M, void* stream)
{
    return launch(false, box1, box2, out, N, M, stream);
}

extern "C" int dh_wh_iou(const float* wh1, const float* wh2, float* out, int64_t N, int64_t M, void* stream)
{
    return launch(true, wh1, wh2, out, N, M, stream);
}
