// YOLOv5 Detect head: the layout change between the 1x1 convolutions and the head maps, fused with the inference decode.
//
// Reference: object_detection/yolov5/models/yolo.py:53-54
//   x[i] = x[i].view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()      (bs, na*no, ny, nx) -> (bs, na, ny, nx, no)
// followed, in eval mode, by the decode of yolo.py:56-70 (see decode.hip).  In the reference the permuted copy is one
// strided pass (read + write of every element) and the decode reads it again; here one launch reads the convolution
// output once and writes both the maps and the decoded tensor (340 B read + 680 B written per anchor instead of
// 680 + 680).  The backward entry is the inverse layout change for the gradient that flows back into the convolutions.
//
// For a fixed (image, anchor) the convolution output is `no` channel rows of ny*nx floats, and the maps hold the same
// numbers as ny*nx rows of `no` floats: a 2-D transpose whose output span is contiguous.  A workgroup owns S spatial
// positions of one (image, anchor): every wave loads whole channel rows (lane = position, 256 contiguous bytes per
// load), writes them into LDS transposed (lane stride = no floats; conflict-free when no is odd, as 85 is) so that LDS
// holds the tile in output order, and after one barrier the tile is streamed out with 16-byte LDS reads and 16-byte
// stores; the decode arithmetic sits between that read and the stores and is the arithmetic of k_yolo_decode.
// A half-precision head (model.half(), object_detection/yolov5/val.py:115,129) keeps fp16 convolution outputs and fp16
// maps: every lane then carries two positions per load (S = 128), the tile is held in fp16, and the decode widens to
// fp32 exactly as dh_yolo_decode_f16 does -- or (HALF) keeps the roundings of such a model's own in-place decode and
// writes the fp16 tensor of dh_yolo_decode_half.
#include "dh_common.h"
#include "dh_decode.h"
#include "../../include/dethead.h"

namespace {

struct head_level {
    const void* conv;    // (bs, na*no, ny, nx): forward input / backward output (fp32 or fp16)
    void* maps;          // (bs, na, ny, nx, no): forward output (may be null) / backward input (same element type)
    uint32_t tiles;      // ceil(ny*nx / S)
    uint32_t blocks;     // na * tiles
    uint32_t hw;         // ny*nx
};

struct head_params {
    decode_level lvl[DH_MAX_LEVELS];   // output geometry and box constants; `src` is not used here
    head_level h[DH_MAX_LEVELS];
    int nl;
    int na;
    dh_fastdiv no;
    uint32_t per_image;
    uint32_t rows_per_image;
    float* out;           // decoded (bs, A, no), or null: layout change only (HALF: an fp16 tensor under the same pointer type)
    float* obj_plane;     // as in dh_yolo_decode
    float* logit_plane;
};

__device__ __forceinline__ int find_level(const head_params& P, uint32_t& blk)
{
    int l = 0;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k) {
        if (l == k && k + 1 < P.nl && blk >= P.h[k].blocks) { blk -= P.h[k].blocks; l = k + 1; }
    }
    return l;
}

typedef _Float16 dh_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 dh_h2 __attribute__((ext_vector_type(2)));

template <typename T> struct vec4_of;
template <> struct vec4_of<float> { typedef dh_f4 type; };
template <> struct vec4_of<_Float16> { typedef dh_h4 type; };

// A lane moves 4 bytes of a channel row per load or store: one fp32 position or two fp16 positions.
template <typename T> __device__ __forceinline__ void load_lane(const T* p, T* v);
template <> __device__ __forceinline__ void load_lane<float>(const float* p, float* v) { v[0] = __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ void load_lane<_Float16>(const _Float16* p, _Float16* v)
{
    const dh_h2 h = __builtin_nontemporal_load(reinterpret_cast<const dh_h2*>(p));
    v[0] = h.x; v[1] = h.y;
}
template <typename T> __device__ __forceinline__ void store_lane(T* p, const T* v);
template <> __device__ __forceinline__ void store_lane<float>(float* p, const float* v) { __builtin_nontemporal_store(v[0], p); }
template <> __device__ __forceinline__ void store_lane<_Float16>(_Float16* p, const _Float16* v)
{
    const dh_h2 h = {v[0], v[1]};
    __builtin_nontemporal_store(h, reinterpret_cast<dh_h2*>(p));
}

// channel rows of the convolution layout <-> LDS tile in map order (tile[s*no + c]); TO_LDS selects the direction.
// The tile spans S = 64 * EPL positions (EPL = positions per lane).  PACKED: ny*nx is a multiple of EPL, so a lane's
// positions are in range together and its 4 bytes are aligned; otherwise (fp16 with an odd ny*nx) they move one by one.
template <typename T, bool TO_LDS, bool PACKED>
__device__ __forceinline__ void channel_rows(T* tile, T* conv_rows, uint32_t no, uint32_t hw, uint32_t ns)
{
    constexpr uint32_t EPL = 4 / sizeof(T);
    const uint32_t wave = threadIdx.x >> 6, s = (threadIdx.x & 63) * EPL;
    for (uint32_t c0 = wave; c0 < no; c0 += 4 * 8) {
        T v[8][EPL];
        if (TO_LDS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {      // all eight loads are issued before the first LDS write
                const uint32_t c = c0 + 4 * u;
                T* row = conv_rows + (size_t)c * hw + s;
                if (PACKED) { if (c < no && s < ns) load_lane<T>(row, v[u]); }
                else {
#pragma unroll
                    for (uint32_t k = 0; k < EPL; ++k) if (c < no && s + k < ns) v[u][k] = row[k];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = c0 + 4 * u;
            if (c >= no) continue;
            T* row = conv_rows + (size_t)c * hw + s;
            if (TO_LDS) {
#pragma unroll
                for (uint32_t k = 0; k < EPL; ++k) if (s + k < ns) tile[(s + k) * no + c] = v[u][k];
            } else {
#pragma unroll
                for (uint32_t k = 0; k < EPL; ++k) if (s + k < ns) v[u][k] = tile[(s + k) * no + c];
                if (PACKED) { if (s < ns) store_lane<T>(row, v[u]); }
                else {
#pragma unroll
                    for (uint32_t k = 0; k < EPL; ++k) if (s + k < ns) row[k] = v[u][k];
                }
            }
        }
    }
}

template <bool HALF> struct decoded_of { typedef float type; typedef dh_f4 type4; };
template <> struct decoded_of<true> { typedef _Float16 type; typedef dh_h4 type4; };

template <typename T, bool VEC, bool PACKED, bool HALF = false>
__global__ void __launch_bounds__(256) k_yolo_head_maps(const head_params P)
{
    constexpr uint32_t S = 64 * (4 / sizeof(T));
    typedef typename vec4_of<T>::type T4;
    typedef typename decoded_of<HALF>::type O;
    typedef typename decoded_of<HALF>::type4 O4;
    extern __shared__ __align__(16) unsigned char tile_bytes[];
    T* tile = reinterpret_cast<T*>(tile_bytes);
    const uint32_t b = blockIdx.y;
    uint32_t blk = blockIdx.x;
    const int l = find_level(P, blk);
    const decode_level& L = P.lvl[l];
    const head_level& H = P.h[l];
    const uint32_t no = P.no.d, hw = H.hw;
    const uint32_t a = blk / H.tiles;
    const uint32_t s0 = (blk - a * H.tiles) * S;
    const uint32_t ns = min(S, hw - s0);
    const size_t plane_row = (size_t)(b * P.na + a) * hw + s0;       // first map row of the tile, (bs, na, ny*nx) order

    channel_rows<T, true, PACKED>(tile, const_cast<T*>(static_cast<const T*>(H.conv)) + (size_t)(b * P.na + a) * no * hw + s0, no, hw, ns);
    __syncthreads();

    T* __restrict__ maps = H.maps ? static_cast<T*>(H.maps) + plane_row * no : nullptr;
    O* __restrict__ dec = P.out ? reinterpret_cast<O*>(P.out) + (size_t)b * P.per_image + L.first + ((size_t)a * hw + s0) * no : nullptr;
    const size_t plane_at = (size_t)b * P.rows_per_image + L.row_first + (size_t)a * hw + s0;
    const float stride = L.stride, aw = L.anchor_px[2 * a], ah = L.anchor_px[2 * a + 1];
    const uint32_t n = ns * no;
    if (VEC) {
        for (uint32_t e = threadIdx.x * 4; e < n; e += 1024) {
            const T4 raw = *reinterpret_cast<const T4*>(tile + e);
            if (maps) __builtin_nontemporal_store(raw, reinterpret_cast<T4*>(maps + e));
            if (!dec) continue;
            const dh_f4 x = {(float)raw.x, (float)raw.y, (float)raw.z, (float)raw.w};
            float y[4] = {dh_sigmoid(x.x), dh_sigmoid(x.y), dh_sigmoid(x.z), dh_sigmoid(x.w)};
            if (HALF) {                                                                        // yolo.py:58 on an fp16 tensor
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = dh_round_f16(y[k]);
            }
            const uint32_t s = dh_div(e, P.no);
            const uint32_t c0 = e - s * no;
            if (c0 < 4 || c0 + 3 >= no) {           // the vector touches box channels of row s or of row s + 1 (never both)
                const bool next = c0 >= 4;
                const uint32_t cell = s0 + s + (next ? 1u : 0u);
                const int k0 = next ? (int)(no - c0) : -(int)c0;
                const uint32_t gy = dh_div(cell, L.nx);
                const uint32_t gx = cell - gy * L.nx.d;
                const float fgx = (float)gx, fgy = (float)gy;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = k - k0;
                    const float sv = y[k];
                    const float t = sv * 2.0f;
                    float xy, wh;
                    if (HALF) {                       // the roundings of decode_box_half
                        xy = dh_round_f16((dh_round_f16(t - 0.5f) + ((c & 1) ? fgy : fgx)) * stride);
                        wh = dh_round_f16(dh_round_f16(t * t) * ((c & 1) ? ah : aw));
                    } else {
                        xy = (sv * 2.0f - 0.5f + ((c & 1) ? fgy : fgx)) * stride;             // yolo.py:60,66
                        wh = (t * t) * ((c & 1) ? ah : aw);                                   // yolo.py:61,67
                    }
                    if (c >= 0 && c < 4) y[k] = c < 2 ? xy : wh;
                }
            }
            if (P.obj_plane) {
                int k4 = 4 - (int)c0;
                uint32_t s4 = s;
                if (k4 < 0) { k4 += (int)no; s4 = s + 1; }
                if (k4 < 4) {
                    P.obj_plane[plane_at + s4] = k4 == 0 ? y[0] : (k4 == 1 ? y[1] : (k4 == 2 ? y[2] : y[3]));
                    if (P.logit_plane) P.logit_plane[plane_at + s4] = k4 == 0 ? x.x : (k4 == 1 ? x.y : (k4 == 2 ? x.z : x.w));
                }
            }
            const O4 o = {(O)y[0], (O)y[1], (O)y[2], (O)y[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<O4*>(dec + e));
        }
    } else {
        for (uint32_t e = threadIdx.x; e < n; e += 256) {
            const T raw = tile[e];
            if (maps) maps[e] = raw;
            if (!dec) continue;
            const float x = (float)raw;
            float y = dh_sigmoid(x);
            if (HALF) y = dh_round_f16(y);
            const uint32_t s = dh_div(e, P.no);
            const uint32_t c = e - s * no;
            if (c < 4) y = HALF ? decode_box_half(L, a * hw + s0 + s, c, y) : decode_box(L, a * hw + s0 + s, c, y);
            if (c == 4 && P.obj_plane) {
                P.obj_plane[plane_at + s] = y;
                if (P.logit_plane) P.logit_plane[plane_at + s] = x;
            }
            dec[e] = (O)y;
        }
    }
}

// gradient w.r.t. the maps (bs, na, ny, nx, no) -> gradient w.r.t. the convolution output (bs, na*no, ny, nx)
template <typename T, bool VEC, bool PACKED>
__global__ void __launch_bounds__(256) k_yolo_head_maps_bwd(const head_params P)
{
    constexpr uint32_t S = 64 * (4 / sizeof(T));
    typedef typename vec4_of<T>::type T4;
    extern __shared__ __align__(16) unsigned char tile_bytes[];
    T* tile = reinterpret_cast<T*>(tile_bytes);
    const uint32_t b = blockIdx.y;
    uint32_t blk = blockIdx.x;
    const int l = find_level(P, blk);
    const head_level& H = P.h[l];
    const uint32_t no = P.no.d, hw = H.hw;
    const uint32_t a = blk / H.tiles;
    const uint32_t s0 = (blk - a * H.tiles) * S;
    const uint32_t ns = min(S, hw - s0);
    const T* __restrict__ g = static_cast<const T*>(H.maps) + ((size_t)(b * P.na + a) * hw + s0) * no;
    const uint32_t n = ns * no;
    if (VEC) {
        for (uint32_t e = threadIdx.x * 4; e < n; e += 1024)
            *reinterpret_cast<T4*>(tile + e) = __builtin_nontemporal_load(reinterpret_cast<const T4*>(g + e));
    } else {
        for (uint32_t e = threadIdx.x; e < n; e += 256) tile[e] = g[e];
    }
    __syncthreads();
    channel_rows<T, false, PACKED>(tile, const_cast<T*>(static_cast<const T*>(H.conv)) + (size_t)(b * P.na + a) * no * hw + s0, no, hw, ns);
}

// fills P and the grid width; returns DH_OK or an error code.  esize: bytes per element of the convolution outputs / maps
int head_setup(head_params& P, const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl,
               int bs, int na, int no, const float* anchor_px, const float* strides, size_t esize, uint32_t* gx, bool* vec,
               bool* packed)
{
    static const float zeros[DH_MAX_LEVELS * 2 * DH_MAX_ANCHORS] = {};
    const uint32_t S = 64 * (uint32_t)(4 / esize);
    if ((size_t)S * no * esize > 65536) return DH_ERR_ARG;       // the tile has to fit the default LDS allowance
    P.nl = nl;
    P.na = na;
    P.no = dh_make_fastdiv((uint32_t)no);
    if (!dh_make_levels(reinterpret_cast<const float* const*>(conv_ptrs), ny, nx, nl, na, no, anchor_px ? anchor_px : zeros,
                        strides ? strides : zeros, P.lvl, &P.per_image, &P.rows_per_image, nullptr))
        return DH_ERR_ARG;
    *gx = 0;
    *vec = true;
    *packed = true;
    for (int l = 0; l < nl; ++l) {
        head_level& H = P.h[l];
        H.conv = conv_ptrs[l];
        H.maps = map_ptrs ? map_ptrs[l] : nullptr;
        H.hw = (uint32_t)(ny[l] * nx[l]);
        if ((uint64_t)bs * na * H.hw >= (1ull << 32)) return DH_ERR_ARG;     // 32-bit row indices
        H.tiles = (H.hw + S - 1) / S;
        H.blocks = (uint32_t)na * H.tiles;
        *gx += H.blocks;
        if (reinterpret_cast<uintptr_t>(H.conv) & 3u) return DH_ERR_ALIGN;
        *packed = *packed && H.hw % (4 / esize) == 0;
        *vec = *vec && ((uint64_t)H.hw * no) % 4 == 0 && (!H.maps || (reinterpret_cast<uintptr_t>(H.maps) & (4 * esize - 1)) == 0);
    }
    return DH_OK;
}

template <typename T, bool HALF = false>
int head_maps_run(const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl, int bs, int na,
                  int no, const float* anchor_px, const float* strides, float* out, float* obj_plane, float* obj_logit_plane,
                  void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || na > DH_MAX_ANCHORS || no < 5 || bs < 0 || bs > 65535) return DH_ERR_ARG;
    if (!conv_ptrs || !ny || !nx || (!map_ptrs && !out)) return DH_ERR_ARG;
    if (out && (!anchor_px || !strides)) return DH_ERR_ARG;
    if ((obj_plane || obj_logit_plane) && !out) return DH_ERR_ARG;
    if (obj_logit_plane && !obj_plane) return DH_ERR_ARG;
    for (int l = 0; map_ptrs && l < nl; ++l)
        if (!map_ptrs[l]) return DH_ERR_ARG;
    head_params P;
    uint32_t gx = 0;
    bool vec = false, packed = false;
    const int rc = head_setup(P, conv_ptrs, map_ptrs, ny, nx, nl, bs, na, no, anchor_px, strides, sizeof(T), &gx, &vec, &packed);
    if (rc != DH_OK) return rc;
    if (bs == 0) return DH_OK;
    P.out = out;
    P.obj_plane = obj_plane;
    P.logit_plane = obj_logit_plane;
    vec = vec && (!out || (HALF ? (reinterpret_cast<uintptr_t>(out) & 7u) == 0 : dh_aligned16(out))) && no >= 7;
    if (HALF && out && (reinterpret_cast<uintptr_t>(out) & 1u)) return DH_ERR_ALIGN;
    const size_t lds = (size_t)64 * (4 / sizeof(T)) * no * sizeof(T);
    hipStream_t st = (hipStream_t)stream;
    if (vec && packed) hipLaunchKernelGGL((k_yolo_head_maps<T, true, true, HALF>), dim3(gx, bs), dim3(256), lds, st, P);
    else if (vec) hipLaunchKernelGGL((k_yolo_head_maps<T, true, false, HALF>), dim3(gx, bs), dim3(256), lds, st, P);
    else if (packed) hipLaunchKernelGGL((k_yolo_head_maps<T, false, true, HALF>), dim3(gx, bs), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_yolo_head_maps<T, false, false, HALF>), dim3(gx, bs), dim3(256), lds, st, P);
    return dh_launch_status();
}

template <typename T>
int head_maps_bwd_run(const void* const* grad_map_ptrs, void* const* grad_conv_ptrs, const int* ny, const int* nx, int nl,
                      int bs, int na, int no, void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || na > DH_MAX_ANCHORS || no < 1 || bs < 0 || bs > 65535) return DH_ERR_ARG;
    if (!grad_map_ptrs || !grad_conv_ptrs || !ny || !nx) return DH_ERR_ARG;
    for (int l = 0; l < nl; ++l)
        if (!grad_map_ptrs[l] || !grad_conv_ptrs[l]) return DH_ERR_ARG;
    head_params P;
    uint32_t gx = 0;
    bool vec = false, packed = false;
    // head_level.conv is the side with channel rows (here the output), head_level.maps the side in map order (here the input)
    const int rc = head_setup(P, const_cast<const void* const*>(grad_conv_ptrs), const_cast<void* const*>(grad_map_ptrs), ny, nx,
                              nl, bs, na, no, nullptr, nullptr, sizeof(T), &gx, &vec, &packed);
    if (rc != DH_OK) return rc;
    if (bs == 0) return DH_OK;
    P.out = P.obj_plane = P.logit_plane = nullptr;
    const size_t lds = (size_t)64 * (4 / sizeof(T)) * no * sizeof(T);
    hipStream_t st = (hipStream_t)stream;
    if (vec && packed) hipLaunchKernelGGL((k_yolo_head_maps_bwd<T, true, true>), dim3(gx, bs), dim3(256), lds, st, P);
    else if (vec) hipLaunchKernelGGL((k_yolo_head_maps_bwd<T, true, false>), dim3(gx, bs), dim3(256), lds, st, P);
    else if (packed) hipLaunchKernelGGL((k_yolo_head_maps_bwd<T, false, true>), dim3(gx, bs), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_yolo_head_maps_bwd<T, false, false>), dim3(gx, bs), dim3(256), lds, st, P);
    return dh_launch_status();
}

}  // namespace

extern "C" int dh_yolo_head_maps(const float* const* conv_ptrs, float* const* map_ptrs, const int* ny, const int* nx, int nl,
                                 int bs, int na, int no, const float* anchor_px, const float* strides, float* out,
                                 float* obj_plane, float* obj_logit_plane, void* stream)
{
    return head_maps_run<float>(reinterpret_cast<const void* const*>(conv_ptrs), reinterpret_cast<void* const*>(map_ptrs), ny, nx,
                                nl, bs, na, no, anchor_px, strides, out, obj_plane, obj_logit_plane, stream);
}

extern "C" int dh_yolo_head_maps_f16(const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl,
                                     int bs, int na, int no, const float* anchor_px, const float* strides, float* out,
                                     float* obj_plane, float* obj_logit_plane, void* stream)
{
    return head_maps_run<_Float16>(conv_ptrs, map_ptrs, ny, nx, nl, bs, na, no, anchor_px, strides, out, obj_plane,
                                   obj_logit_plane, stream);
}

extern "C" int dh_yolo_head_maps_half(const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl,
                                      int bs, int na, int no, const float* anchor_px, const float* strides, void* out_f16,
                                      float* obj_plane, void* stream)
{
    if (!out_f16) return DH_ERR_ARG;
    return head_maps_run<_Float16, true>(conv_ptrs, map_ptrs, ny, nx, nl, bs, na, no, anchor_px, strides,
                                         reinterpret_cast<float*>(out_f16), obj_plane, nullptr, stream);
}

extern "C" int dh_yolo_head_maps_bwd(const float* const* grad_map_ptrs, float* const* grad_conv_ptrs, const int* ny,
                                     const int* nx, int nl, int bs, int na, int no, void* stream)
{
    return head_maps_bwd_run<float>(reinterpret_cast<const void* const*>(grad_map_ptrs),
                                    reinterpret_cast<void* const*>(grad_conv_ptrs), ny, nx, nl, bs, na, no, stream);
}

extern "C" int dh_yolo_head_maps_bwd_f16(const void* const* grad_map_ptrs, void* const* grad_conv_ptrs, const int* ny,
                                         const int* nx, int nl, int bs, int na, int no, void* stream)
{
    return head_maps_bwd_run<_Float16>(grad_map_ptrs, grad_conv_ptrs, ny, nx, nl, bs, na, no, stream);
}
