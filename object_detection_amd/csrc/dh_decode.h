// Per-level record of the YOLOv5 Detect head and the box arithmetic of its inference branch
// (object_detection/yolov5/models/yolo.py:56-70,72-81), shared by the decode kernel and the fused decode+NMS path.
#pragma once
#include "dh_common.h"
#include "../../include/dethead.h"

struct decode_level {
    const float* src;     // (bs, na, ny, nx, no) contiguous; fp16 maps travel under the same pointer type and are
                          // cast back by the kernels that were told so
    uint32_t first;       // first output element of this level inside one image's span
    uint32_t row_first;   // first output row (anchor) of this level inside one image
    uint32_t elems;       // na*ny*nx*no
    dh_fastdiv plane;     // ny*nx
    dh_fastdiv nx;
    float stride;
    float anchor_px[2 * DH_MAX_ANCHORS];   // anchors[l][a] * stride[l]  (w,h)
};

// box channels (c < 4) of row `row`: xy = (2s - 0.5 + grid) * stride, wh = (2s)^2 * anchor_px
__device__ __forceinline__ float decode_box(const decode_level& L, uint32_t row, uint32_t c, float s)
{
    const uint32_t a = dh_div(row, L.plane);
    const uint32_t cell = row - a * L.plane.d;
    const uint32_t gy = dh_div(cell, L.nx);
    const uint32_t gx = cell - gy * L.nx.d;
    const float g = (float)((c & 1u) ? gy : gx);
    const float xy = (s * 2.0f - 0.5f + g) * L.stride;
    const float t = s * 2.0f;
    const float wh = (t * t) * L.anchor_px[2 * a + (c & 1u)];
    return c < 2 ? xy : wh;
}

// The same behind a half-precision model (model.half(), val.py:129-130; yolo.py:59-61 in place): the maps and the sigmoid
// are fp16, grid, stride and anchor grid are float tensors, so torch rounds to fp16 after the sigmoid (s comes in rounded),
// after *2 (exact), after -0.5 and after **2, widens for + grid, * stride and * anchor_grid, and rounds once more when the
// result is assigned into the fp16 tensor.  anchor_px must be what such a model holds: (anchors.half() * stride).float().
__device__ __forceinline__ float dh_round_f16(float v) { return (float)(_Float16)v; }

__device__ __forceinline__ float decode_box_half(const decode_level& L, uint32_t row, uint32_t c, float s)
{
    const uint32_t a = dh_div(row, L.plane);
    const uint32_t cell = row - a * L.plane.d;
    const uint32_t gy = dh_div(cell, L.nx);
    const uint32_t gx = cell - gy * L.nx.d;
    const float g = (float)((c & 1u) ? gy : gx);
    const float t = s * 2.0f;
    const float xy = (dh_round_f16(t - 0.5f) + g) * L.stride;
    const float wh = dh_round_f16(t * t) * L.anchor_px[2 * a + (c & 1u)];
    return dh_round_f16(c < 2 ? xy : wh);
}

// Fills the level records from the host-side description; returns false on an unsupported shape.
// vec4_ok (optional) reports whether every level span is a multiple of 4 elements and aligned for a 4-element load
// (align_mask 15 for fp32 maps, 7 for fp16 maps).
static inline bool dh_make_levels(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int na, int no,
                                  const float* anchor_px, const float* strides, decode_level* lvl, uint32_t* per_image,
                                  uint32_t* rows_per_image, bool* vec4_ok, uintptr_t align_mask = 15)
{
    uint64_t first = 0;
    uint32_t row_first = 0;
    bool vec4 = true;
    for (int l = 0; l < nl; ++l) {
        if (!level_ptrs[l] || ny[l] < 1 || nx[l] < 1) return false;
        const uint64_t elems = (uint64_t)na * ny[l] * nx[l] * no;
        if (first + elems >= (1ull << 31)) return false;   // 32-bit element indices inside one image
        decode_level& L = lvl[l];
        L.src = level_ptrs[l];
        L.first = (uint32_t)first;
        L.row_first = row_first;
        row_first += (uint32_t)(na * ny[l] * nx[l]);
        L.elems = (uint32_t)elems;
        L.plane = dh_make_fastdiv((uint32_t)(ny[l] * nx[l]));
        L.nx = dh_make_fastdiv((uint32_t)nx[l]);
        L.stride = strides[l];
        for (int k = 0; k < 2 * na; ++k) L.anchor_px[k] = anchor_px[l * 2 * na + k];
        vec4 = vec4 && (elems % 4 == 0) && (reinterpret_cast<uintptr_t>(level_ptrs[l]) & align_mask) == 0;
        first += elems;
    }
    *per_image = (uint32_t)first;
    *rows_per_image = row_first;
    if (vec4_ok) *vec4_ok = vec4;
    return true;
}
