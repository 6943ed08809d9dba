// YOLOv5 Detect head, inference branch: sigmoid + grid/anchor decode of all levels in one launch.
//
// Reference: object_detection/yolov5/models/yolo.py:56-70 (decode, concat) and :72-81 (_make_grid).
//   y = sigmoid(x);  xy = (2y - 0.5 + (gx,gy)) * stride;  wh = (2y)^2 * (anchor*stride);  rest = y
//   output row order: level, anchor, gy, gx  -> (bs, sum_l na*ny*nx, no)
// Per image the output is one contiguous span and each level's input is one contiguous span, so
// the kernel is a pure stream: 16-byte loads and stores, 680 B of HBM traffic per anchor, and the
// reference's separate sigmoid / slice-assign / torch.cat passes collapse into this single pass.
#include <stdlib.h>

#include "dh_common.h"
#include "dh_decode.h"
#include "../../include/dethead.h"

namespace {

struct decode_params {
    decode_level lvl[DH_MAX_LEVELS];
    uint32_t blocks[DH_MAX_LEVELS];   // blocks assigned to each level along grid x
    int nl;
    dh_fastdiv no;
    uint32_t per_image;   // elements of one image's output span = A*no
    uint32_t rows_per_image;
    float* obj_plane;     // optional (bs, A) copy of the objectness column, or nullptr
    float* logit_plane;   // optional (bs, A) copy of the raw objectness logit (input channel 4), or nullptr
};

// One block owns DH_DEC_UNROLL*256 consecutive vectors of ONE level of ONE image, so the level
// record is wave-uniform (scalar registers).  VEC = 4 needs every level span to be a multiple of
// 4 floats and 16-byte aligned, and no >= 7 (a vector then touches the box channels of at most one
// row); VEC = 1 is the general fallback.  Every element gets its sigmoid unconditionally; a vector
// that holds box channels patches them afterwards.  A wave's 256 consecutive floats always contain
// some row's box, so that patch runs in every wave for a few lanes: it is kept short (the row is
// decomposed once per vector, not once per element) and free of memory instructions (the level
// record is read into scalar registers up front), so the loads already in flight are never waited for.
#define DH_PIN_SGPR(v) asm volatile("" : "+s"(v))
typedef _Float16 dh_h4 __attribute__((ext_vector_type(4)));

// four consecutive map elements as fp32; fp16 maps (T = _Float16) are widened on load, and everything after the load is
// the fp32 arithmetic: a half-precision head decodes to exactly what its maps converted with .float() decode to
template <bool NT_LOAD>
__device__ __forceinline__ dh_f4 load4(const float* p)
{
    if (NT_LOAD) return __builtin_nontemporal_load(reinterpret_cast<const dh_f4*>(p));
    return *reinterpret_cast<const dh_f4*>(p);
}
template <bool NT_LOAD>
__device__ __forceinline__ dh_f4 load4(const _Float16* p)
{
    const dh_h4 h = NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const dh_h4*>(p)) : *reinterpret_cast<const dh_h4*>(p);
    const dh_f4 f = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return f;
}

template <int VEC, int DH_DEC_UNROLL, int NT, typename T = float>
__global__ void __launch_bounds__(256) k_yolo_decode(const decode_params P, float* __restrict__ out)
{
    const int b = blockIdx.y;
    int l = 0;
    uint32_t blk = blockIdx.x;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k) {
        if (l == k && k + 1 < P.nl && blk >= P.blocks[k]) { blk -= P.blocks[k]; l = k + 1; }
    }
    const decode_level& L = P.lvl[l];
    const uint32_t nvec = L.elems / VEC;
    const uint32_t no = P.no.d;
    const T* __restrict__ src = reinterpret_cast<const T*>(L.src) + (size_t)b * L.elems;
    float* __restrict__ dst = out + (size_t)b * P.per_image + L.first;
    const uint32_t v0 = blk * (256 * DH_DEC_UNROLL) + threadIdx.x;
    if (VEC == 4) {
        dh_fastdiv plane = L.plane, nxd = L.nx;
        uint32_t stride_bits = __float_as_uint(L.stride);
        uint32_t anc[2 * DH_MAX_ANCHORS];
#pragma unroll
        for (int q = 0; q < 2 * DH_MAX_ANCHORS; ++q) { anc[q] = __float_as_uint(L.anchor_px[q]); DH_PIN_SGPR(anc[q]); }
        DH_PIN_SGPR(plane.magic); DH_PIN_SGPR(plane.shift); DH_PIN_SGPR(plane.d);
        DH_PIN_SGPR(nxd.magic); DH_PIN_SGPR(nxd.shift); DH_PIN_SGPR(nxd.d); DH_PIN_SGPR(stride_bits);
        const float stride = __uint_as_float(stride_bits);
        dh_f4 x[DH_DEC_UNROLL];
#pragma unroll
        for (int u = 0; u < DH_DEC_UNROLL; ++u) {
            const uint32_t v = v0 + u * 256;
            if (v < nvec) x[u] = load4<(NT & 1) != 0>(src + (size_t)v * 4);
        }
#pragma unroll
        for (int u = 0; u < DH_DEC_UNROLL; ++u) {
            const uint32_t v = v0 + u * 256;
            if (v >= nvec) continue;
            const uint32_t e = v * 4;
            float y[4] = {dh_sigmoid(x[u].x), dh_sigmoid(x[u].y), dh_sigmoid(x[u].z), dh_sigmoid(x[u].w)};
            const uint32_t row = dh_div(e, P.no);
            const uint32_t c0 = e - row * no;
            if (c0 < 4 || c0 + 3 >= no) {           // the vector touches box channels of `row` or of `row + 1` (never both)
                const bool next = c0 >= 4;
                const uint32_t r = row + (next ? 1u : 0u);
                const int k0 = next ? (int)(no - c0) : -(int)c0;      // element of this vector where channel 0 of row r sits
                const uint32_t a = dh_div(r, plane);
                const uint32_t cell = r - a * plane.d;
                const uint32_t gy = dh_div(cell, nxd);
                const uint32_t gx = cell - gy * nxd.d;
                uint32_t aw = anc[0], ah = anc[1];
#pragma unroll
                for (int q = 1; q < DH_MAX_ANCHORS; ++q)
                    if (a == (uint32_t)q) { aw = anc[2 * q]; ah = anc[2 * q + 1]; }
                const float fgx = (float)gx, fgy = (float)gy;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = k - k0;
                    const float sv = y[k];
                    const float xy = (sv * 2.0f - 0.5f + ((c & 1) ? fgy : fgx)) * stride;     // yolo.py:60,66
                    const float t = sv * 2.0f;
                    const float wh = (t * t) * __uint_as_float((c & 1) ? ah : aw);            // yolo.py:61,67
                    if (c >= 0 && c < 4) y[k] = c < 2 ? xy : wh;
                }
            }
            if (P.obj_plane) {       // channel 4 of `row` sits at offset 4 - c0, of `row + 1` at 4 - c0 + no
                int k4 = 4 - (int)c0;
                uint32_t r4 = row;
                if (k4 < 0) { k4 += (int)no; r4 = row + 1; }
                if (k4 < 4) {
                    const float ov = k4 == 0 ? y[0] : (k4 == 1 ? y[1] : (k4 == 2 ? y[2] : y[3]));
                    const size_t at = (size_t)b * P.rows_per_image + L.row_first + r4;
                    P.obj_plane[at] = ov;
                    if (P.logit_plane) P.logit_plane[at] = k4 == 0 ? x[u].x : (k4 == 1 ? x[u].y : (k4 == 2 ? x[u].z : x[u].w));
                }
            }
            const dh_f4 o = {y[0], y[1], y[2], y[3]};
            if (NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<dh_f4*>(dst + (size_t)e));
            else *reinterpret_cast<dh_f4*>(dst + (size_t)e) = o;
        }
    } else {
#pragma unroll
        for (int u = 0; u < DH_DEC_UNROLL; ++u) {
            const uint32_t v = v0 + u * 256;
            if (v >= nvec) continue;
            const float xv = (float)src[v];
            float y = dh_sigmoid(xv);
            const uint32_t row = dh_div(v, P.no);
            const uint32_t c = v - row * no;
            if (c < 4) y = decode_box(L, row, c, y);
            if (c == 4 && P.obj_plane) {
                P.obj_plane[(size_t)b * P.rows_per_image + L.row_first + row] = y;
                if (P.logit_plane) P.logit_plane[(size_t)b * P.rows_per_image + L.row_first + row] = xv;
            }
            dst[v] = y;
        }
    }
}

// The same decode with the box patch moved off the streaming lanes.  A wave's two spans of 256 consecutive floats hold the
// box and objectness channels of at most `specials / 10` rows each; instead of every lane running the patch code over the four
// components of its vectors (select chains over register components, executed by the whole wave for the few lanes that
// need it), the wave writes its sigmoids (and logits) to a slice of LDS, lane s takes the s-th of those few special elements
// -- (span, row, channel 0..4) -- patches it in LDS or copies it to the objectness planes, and every lane then reads its
// vectors back and stores them.  Same arithmetic, element for element; no workgroup barrier (the slice is the wave's own).
__device__ __forceinline__ void dh_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int U, int NT, typename T = float>
__global__ void __launch_bounds__(256) k_yolo_decode_lanes(const decode_params P, float* __restrict__ out, const uint32_t specials)
{
    __shared__ __align__(16) float s_y[4 * U * 256];
    __shared__ __align__(16) float s_x[4 * U * 256];
    const int b = blockIdx.y;
    int l = 0;
    uint32_t blk = blockIdx.x;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k) {
        if (l == k && k + 1 < P.nl && blk >= P.blocks[k]) { blk -= P.blocks[k]; l = k + 1; }
    }
    const decode_level& L = P.lvl[l];
    const uint32_t nvec = L.elems / 4;
    const uint32_t no = P.no.d;
    const T* __restrict__ src = reinterpret_cast<const T*>(L.src) + (size_t)b * L.elems;
    float* __restrict__ dst = out + (size_t)b * P.per_image + L.first;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v0 = blk * (256 * U) + threadIdx.x;
    dh_fastdiv plane = L.plane, nxd = L.nx;
    uint32_t stride_bits = __float_as_uint(L.stride);
    uint32_t anc[2 * DH_MAX_ANCHORS];
#pragma unroll
    for (int q = 0; q < 2 * DH_MAX_ANCHORS; ++q) { anc[q] = __float_as_uint(L.anchor_px[q]); DH_PIN_SGPR(anc[q]); }
    DH_PIN_SGPR(plane.magic); DH_PIN_SGPR(plane.shift); DH_PIN_SGPR(plane.d);
    DH_PIN_SGPR(nxd.magic); DH_PIN_SGPR(nxd.shift); DH_PIN_SGPR(nxd.d); DH_PIN_SGPR(stride_bits);
    const float stride = __uint_as_float(stride_bits);
    dh_f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t v = v0 + u * 256;
        const dh_f4 zero = {0.f, 0.f, 0.f, 0.f};
        x[u] = v < nvec ? load4<(NT & 1) != 0>(src + (size_t)v * 4) : zero;
    }
    dh_f4* __restrict__ wy = reinterpret_cast<dh_f4*>(s_y + wave * (U * 256));
    dh_f4* __restrict__ wx = reinterpret_cast<dh_f4*>(s_x + wave * (U * 256));
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const dh_f4 y = {dh_sigmoid(x[u].x), dh_sigmoid(x[u].y), dh_sigmoid(x[u].z), dh_sigmoid(x[u].w)};
        wy[u * 64 + lane] = y;
        if (P.logit_plane) wx[u * 64 + lane] = x[u];
    }
    dh_wave_sync();
    const uint32_t per_span = specials / U;
    const uint32_t vw = blk * (256 * U) + wave * 64;           // first vector of the wave's first span
    float* __restrict__ fy = s_y + wave * (U * 256);
    const float* __restrict__ fx = s_x + wave * (U * 256);
    for (uint32_t s = lane; s < specials; s += 64) {
        uint32_t u = 0;                                        // s / per_span without a division (U <= 4)
#pragma unroll
        for (int q = 1; q < U; ++q) u += s >= (uint32_t)q * per_span ? 1u : 0u;
        const uint32_t t = s - u * per_span;
        const uint32_t ri = t / 5u, c = t - ri * 5u;
        const uint32_t e0 = (vw + u * 256) * 4;                // first element of the span
        const uint32_t r = dh_div(e0, P.no) + ri;
        const uint32_t e = r * no + c;
        if (e < e0 || e >= e0 + 256 || e >= L.elems) continue;
        const uint32_t at = u * 256 + (e - e0);
        const float sv = fy[at];
        if (c < 4) {
            const uint32_t a = dh_div(r, plane);
            const uint32_t cell = r - a * plane.d;
            const uint32_t gy = dh_div(cell, nxd);
            const uint32_t gx = cell - gy * nxd.d;
            uint32_t aw = anc[0], ah = anc[1];
#pragma unroll
            for (int q = 1; q < DH_MAX_ANCHORS; ++q)
                if (a == (uint32_t)q) { aw = anc[2 * q]; ah = anc[2 * q + 1]; }
            const float xy = (sv * 2.0f - 0.5f + (float)((c & 1) ? gy : gx)) * stride;          // yolo.py:60,66
            const float t2 = sv * 2.0f;
            const float wh = (t2 * t2) * __uint_as_float((c & 1) ? ah : aw);                     // yolo.py:61,67
            fy[at] = c < 2 ? xy : wh;
        } else if (P.obj_plane) {
            const size_t row_at = (size_t)b * P.rows_per_image + L.row_first + r;
            P.obj_plane[row_at] = sv;
            if (P.logit_plane) P.logit_plane[row_at] = fx[at];
        }
    }
    dh_wave_sync();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t v = v0 + u * 256;
        if (v >= nvec) continue;
        const dh_f4 o = wy[u * 64 + lane];
        if (NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<dh_f4*>(dst + (size_t)v * 4));
        else *reinterpret_cast<dh_f4*>(dst + (size_t)v * 4) = o;
    }
}

}  // namespace

namespace {

int decode_run(const float* const* level_ptrs, bool half, const int* ny, const int* nx, int nl, int bs, int na, int no,
               const float* anchor_px, const float* strides, float* out, float* obj_plane, float* obj_logit_plane, void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || na > DH_MAX_ANCHORS || no < 5 || bs < 0 || bs > 65535) return DH_ERR_ARG;
    if (!level_ptrs || !ny || !nx || !anchor_px || !strides || (bs > 0 && !out)) return DH_ERR_ARG;
    if (bs == 0) return DH_OK;
    decode_params P;
    P.nl = nl;
    P.no = dh_make_fastdiv((uint32_t)no);
    bool vec4 = false;
    if (!dh_make_levels(level_ptrs, ny, nx, nl, na, no, anchor_px, strides, P.lvl, &P.per_image, &P.rows_per_image, &vec4,
                        half ? 7 : 15))
        return DH_ERR_ARG;
    for (int l = 0; half && l < nl; ++l)
        if (reinterpret_cast<uintptr_t>(level_ptrs[l]) & 1u) return DH_ERR_ALIGN;
    vec4 = vec4 && dh_aligned16(out) && no >= 7;
    if (obj_logit_plane && !obj_plane) return DH_ERR_ARG;   // the logit copy rides on the objectness-plane path
    P.obj_plane = obj_plane;
    P.logit_plane = obj_logit_plane;
    const int vec = vec4 ? 4 : 1;
    // tuning knobs, read once (defaults are the measured best): unroll factor and non-temporal policy (bit 0 loads, bit 1 stores)
    static const int unroll = [] {
        const char* u = getenv("DH_DECODE_UNROLL");
        const int v = u ? atoi(u) : 2;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 2;
    }();
    static const int nt = [] {
        const char* n = getenv("DH_DECODE_NT");
        return n ? atoi(n) : 3;
    }();
    const int per_block = 256 * (vec4 ? unroll : 4);
    uint32_t gx = 0;
    for (int l = 0; l < nl; ++l) {
        const uint32_t nvec = P.lvl[l].elems / vec;
        P.blocks[l] = (nvec + per_block - 1) / per_block;
        gx += P.blocks[l];
    }
    hipStream_t st = (hipStream_t)stream;
    static const int lanes = [] {      // 0: the box patch on the streaming lanes (k_yolo_decode), the form measured slower
        const char* n = getenv("DH_DECODE_LANES");
        return n ? atoi(n) : 1;
    }();
    if (vec4 && (unroll == 1 || unroll == 2 || unroll == 4) && (half ? lanes == 2 : lanes != 0)) {      // fp16 maps: measured slower (0.194 against 0.176 ms), off
        // special elements of a wave: `unroll` spans, each touching at most (no + 254) / no + 1 rows, five channels per row
        const uint32_t specials = (uint32_t)unroll * 5u * (uint32_t)((no + 254) / no + 1);
#define DH_LANES(U) do { if (half) hipLaunchKernelGGL((k_yolo_decode_lanes<U, 3, _Float16>), dim3(gx, bs), dim3(256), 0, st, P, out, specials); \
                         else hipLaunchKernelGGL((k_yolo_decode_lanes<U, 3>), dim3(gx, bs), dim3(256), 0, st, P, out, specials); } while (0)
        if (unroll == 1) DH_LANES(1); else if (unroll == 2) DH_LANES(2); else DH_LANES(4);
#undef DH_LANES
        return dh_launch_status();
    }
#define DH_DEC(U, N) do { if (half) hipLaunchKernelGGL((k_yolo_decode<4, U, N, _Float16>), dim3(gx, bs), dim3(256), 0, st, P, out); \
                          else hipLaunchKernelGGL((k_yolo_decode<4, U, N>), dim3(gx, bs), dim3(256), 0, st, P, out); } while (0)
#define DH_DEC_NT(U) do { if (nt == 0) DH_DEC(U, 0); else if (nt == 1) DH_DEC(U, 1); else if (nt == 2) DH_DEC(U, 2); else DH_DEC(U, 3); } while (0)
    if (vec4) {
        if (unroll == 1) DH_DEC_NT(1); else if (unroll == 2) DH_DEC_NT(2); else if (unroll == 8) DH_DEC_NT(8); else DH_DEC_NT(4);
    } else if (half) {
        hipLaunchKernelGGL((k_yolo_decode<1, 4, 0, _Float16>), dim3(gx, bs), dim3(256), 0, st, P, out);
    } else {
        hipLaunchKernelGGL((k_yolo_decode<1, 4, 0>), dim3(gx, bs), dim3(256), 0, st, P, out);
    }
#undef DH_DEC_NT
#undef DH_DEC
    return dh_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------ half-precision model
// Behind model.half() (val.py:129-130) the head's maps, its sigmoid and the decoded tensor are fp16, while the grid and the
// anchor grid are float tensors (yolo.py:78-80): torch rounds to fp16 after the sigmoid, after *2 (exact), after -0.5 and
// after **2, widens for + grid, * stride and * anchor_grid, and rounds once more when the result is assigned into the fp16
// tensor (yolo.py:59-61).  This kernel performs exactly those roundings and writes fp16.  anchor_px must be what
// _make_grid holds on such a model: (anchors.half() * stride).float().
namespace {

// one thread: VEC consecutive elements of one level of one image (VEC = 4: one 8-byte load).  OUT = _Float16: the in-place
// form (yolo.py:59-61), every value rounded into the fp16 tensor.  OUT = float: the torch.cat form (yolo.py:63-65), where
// xy and wh stay the fp32 values they were computed as and the cat widens the fp16 rest.
template <int VEC, typename OUT, bool ALIGNED>
__global__ void __launch_bounds__(256) k_yolo_decode_half(const decode_params P, OUT* __restrict__ out)
{
    const int b = blockIdx.y;
    int l = 0;
    uint32_t blk = blockIdx.x;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k) {
        if (l == k && k + 1 < P.nl && blk >= P.blocks[k]) { blk -= P.blocks[k]; l = k + 1; }
    }
    const decode_level& L = P.lvl[l];
    const uint32_t nvec = L.elems / VEC;
    const uint32_t no = P.no.d;
    const uint32_t v = blk * 256 + threadIdx.x;
    if (v >= nvec) return;
    const _Float16* __restrict__ src = reinterpret_cast<const _Float16*>(L.src) + (size_t)b * L.elems + (size_t)v * VEC;
    OUT* __restrict__ dst = out + (size_t)b * P.per_image + L.first + (size_t)v * VEC;
    _Float16 x[VEC];
    OUT y[VEC];
    typedef _Float16 dh_h8 __attribute__((ext_vector_type(8)));
    if (VEC == 8) *reinterpret_cast<dh_h8*>(x) = __builtin_nontemporal_load(reinterpret_cast<const dh_h8*>(src));
    else if (VEC == 4) *reinterpret_cast<dh_h4*>(x) = __builtin_nontemporal_load(reinterpret_cast<const dh_h4*>(src));
    else x[0] = src[0];
    const uint32_t row = dh_div(v * VEC, P.no), c0 = v * VEC - row * no;
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = dh_round_f16(dh_sigmoid((float)x[k]));        // yolo.py:58
    // With no >= VEC + 3 (the host's condition for a vector) it holds box channels of at most one row: `row`, or `row + 1`
    // when it runs over the row's end.  That row is decomposed once, not once per element.
    if (c0 < 4 || c0 + VEC - 1 >= no) {
        const bool next = c0 >= 4;
        const uint32_t r = row + (next ? 1u : 0u);
        const int k0 = next ? (int)(no - c0) : -(int)c0;                        // element of this vector where channel 0 of row r sits
        const uint32_t a = dh_div(r, L.plane);
        const uint32_t cell = r - a * L.plane.d;
        const uint32_t gy = dh_div(cell, L.nx);
        const uint32_t gx = cell - gy * L.nx.d;
        float aw = L.anchor_px[0], ah = L.anchor_px[1];
#pragma unroll
        for (int q = 1; q < DH_MAX_ANCHORS; ++q)
            if (a == (uint32_t)q) { aw = L.anchor_px[2 * q]; ah = L.anchor_px[2 * q + 1]; }
        const float fgx = (float)gx, fgy = (float)gy;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = k - k0;
            const float t = s[k] * 2.0f;                                               // exact in fp16
            const float xy = (dh_round_f16(t - 0.5f) + ((c & 1) ? fgy : fgx)) * L.stride;   // yolo.py:60
            const float wh = dh_round_f16(t * t) * ((c & 1) ? ah : aw);                 // yolo.py:61
            if (c >= 0 && c < 4) s[k] = c < 2 ? xy : wh;
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) y[k] = (OUT)s[k];                            // OUT = _Float16: the assignment into the fp16 tensor
    if (P.obj_plane) {           // channel 4 of `row` sits at element 4 - c0 of the vector, of `row + 1` at 4 - c0 + no
        int k4 = 4 - (int)c0;
        uint32_t r4 = row;
        if (k4 < 0) { k4 += (int)no; r4 = row + 1; }
        if (k4 < VEC) {
            float ov = s[0];
#pragma unroll
            for (int k = 1; k < VEC; ++k) if (k4 == k) ov = s[k];
            P.obj_plane[(size_t)b * P.rows_per_image + L.row_first + r4] = ov;
        }
    }
    if (VEC == 8 && ALIGNED) {                                                 // fp16 only: one 16-byte store
        typedef OUT out8 __attribute__((ext_vector_type(8)));
        const out8 y8 = {y[0], y[1], y[2], y[3], y[4 % VEC], y[5 % VEC], y[6 % VEC], y[7 % VEC]};
        __builtin_nontemporal_store(y8, reinterpret_cast<out8*>(dst));
    } else if (VEC == 4 && ALIGNED) {                                          // one 8-byte (fp16) or 16-byte (fp32) store
        typedef OUT out4 __attribute__((ext_vector_type(4)));
        const out4 y4 = {y[0], y[1], y[2], y[3 % VEC]};
        __builtin_nontemporal_store(y4, reinterpret_cast<out4*>(dst));
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) dst[k] = y[k];
    }
}

}  // namespace

extern "C" int dh_yolo_decode_half(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                   const float* anchor_px, const float* strides, int inplace, void* out, float* obj_plane,
                                   void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || na > DH_MAX_ANCHORS || no < 5 || bs < 0 || bs > 65535) return DH_ERR_ARG;
    if (!level_ptrs || !ny || !nx || !anchor_px || !strides || (bs > 0 && !out)) return DH_ERR_ARG;
    if (bs == 0) return DH_OK;
    decode_params P;
    P.nl = nl;
    P.no = dh_make_fastdiv((uint32_t)no);
    bool vec4 = false;
    if (!dh_make_levels(reinterpret_cast<const float* const*>(level_ptrs), ny, nx, nl, na, no, anchor_px, strides, P.lvl, &P.per_image,
                        &P.rows_per_image, &vec4, 7))
        return DH_ERR_ARG;
    for (int l = 0; l < nl; ++l)
        if (reinterpret_cast<uintptr_t>(level_ptrs[l]) & 1u) return DH_ERR_ALIGN;
    if (reinterpret_cast<uintptr_t>(out) & (inplace ? 1u : 3u)) return DH_ERR_ALIGN;
    P.obj_plane = obj_plane;
    P.logit_plane = nullptr;
    // a vector of V elements holds the box channels of at most one row when no >= V + 3; its loads need the level spans, its
    // single store the output spans, to start on multiples of V elements of aligned tensors
    vec4 = vec4 && no >= 7;
    bool st4 = vec4 && P.per_image % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & (inplace ? 7u : 15u)) == 0;
    for (int l = 0; st4 && l < nl; ++l) st4 = P.lvl[l].first % 4 == 0;
    bool v8 = inplace && st4 && no >= 11 && P.per_image % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    for (int l = 0; v8 && l < nl; ++l)
        v8 = P.lvl[l].elems % 8 == 0 && P.lvl[l].first % 8 == 0 && (reinterpret_cast<uintptr_t>(level_ptrs[l]) & 15u) == 0;
    const int vec = v8 ? 8 : (vec4 ? 4 : 1);
    uint32_t gx = 0;
    for (int l = 0; l < nl; ++l) {
        P.blocks[l] = (P.lvl[l].elems / vec + 255) / 256;
        gx += P.blocks[l];
    }
    hipStream_t st = (hipStream_t)stream;
#define DH_DEC_HALF(V, O, A) hipLaunchKernelGGL((k_yolo_decode_half<V, O, A>), dim3(gx, bs), dim3(256), 0, st, P, (O*)out)
    if (v8) DH_DEC_HALF(8, _Float16, true);
    else if (inplace) { if (st4) DH_DEC_HALF(4, _Float16, true); else if (vec4) DH_DEC_HALF(4, _Float16, false); else DH_DEC_HALF(1, _Float16, false); }
    else { if (st4) DH_DEC_HALF(4, float, true); else if (vec4) DH_DEC_HALF(4, float, false); else DH_DEC_HALF(1, float, false); }
#undef DH_DEC_HALF
    return dh_launch_status();
}

extern "C" int dh_yolo_decode(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na,
                              int no, const float* anchor_px, const float* strides, float* out, float* obj_plane,
                              float* obj_logit_plane, void* stream)
{
    return decode_run(level_ptrs, false, ny, nx, nl, bs, na, no, anchor_px, strides, out, obj_plane, obj_logit_plane, stream);
}

extern "C" int dh_yolo_decode_f16(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na,
                                  int no, const float* anchor_px, const float* strides, float* out, float* obj_plane,
                                  float* obj_logit_plane, void* stream)
{
    return decode_run(reinterpret_cast<const float* const*>(level_ptrs), true, ny, nx, nl, bs, na, no, anchor_px, strides, out,
                      obj_plane, obj_logit_plane, stream);
}
