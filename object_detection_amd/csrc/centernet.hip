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

// per cell: conf = max_c tanh(pool_nms(heat)[c]), cls = first argmax
__global__ void __launch_bounds__(256) k_cn_best_class(const float* __restrict__ heat, int C, int W, int H,
                                                       float* __restrict__ conf_out, int* __restrict__ cls_out)
{
    const int b = blockIdx.y;
    const int cells = W * H;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const int x = cell / H, y = cell - x * H;
    const float* __restrict__ img = heat + (size_t)b * C * cells;
    float best = -INFINITY;
    int bc = 0;
    for (int c = 0; c < C; ++c) {
        const float t = tanhf(peak_value(img + (size_t)c * cells, W, H, x, y, 1));
        if (t > best || c == 0) { best = t; bc = c; }     // strict > keeps the first maximum
    }
    conf_out[(size_t)b * cells + cell] = best;
    cls_out[(size_t)b * cells + cell] = bc;
}

// order-preserving compaction of the cells above the confidence threshold, one workgroup per image
__global__ void __launch_bounds__(1024) k_cn_emit(const float* __restrict__ conf, const int* __restrict__ cls,
                                                  const float* __restrict__ size, const float* __restrict__ offset,
                                                  int W, int H, float confidence, float* __restrict__ det,
                                                  int* __restrict__ counts)
{
    __shared__ int s_w[16];
    const int b = blockIdx.x;
    const int cells = W * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float fw = (float)W, fh = (float)H;
    int total = 0;
    for (int base = 0; base < cells; base += 1024) {
        const int cell = base + threadIdx.x;
        float cf = 0.0f;
        bool take = false;
        if (cell < cells) {
            cf = conf[(size_t)b * cells + cell];
            take = cf > confidence;                                                   // dataset.py:189
        }
        const unsigned long long bm = __ballot(take);
        if (lane == 0) s_w[wave] = __popcll(bm);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = s_w[w]; if (w < wave) before += c; all += c; }
        __syncthreads();
        if (take) {
            const int pos = total + before + __popcll(bm & ((1ull << lane) - 1ull));
            const int x = cell / H, y = cell - x * H;
            const float* __restrict__ sz = size + (size_t)b * 2 * cells;
            const float* __restrict__ of = offset + (size_t)b * 2 * cells;
            const float cx = (float)x + of[cell], cy = (float)y + of[cells + cell];    // dataset.py:197-198
            const float hw = sz[cell] / 2, hh = sz[cells + cell] / 2;
            float* __restrict__ d = det + ((size_t)b * cells + pos) * 6;
            d[0] = __fdiv_rn(cx - hw, fw);                                            // dataset.py:200-205
            d[1] = __fdiv_rn(cy - hh, fh);
            d[2] = __fdiv_rn(cx + hw, fw);
            d[3] = __fdiv_rn(cy + hh, fh);
            d[4] = cf;
            d[5] = (float)cls[(size_t)b * cells + cell];
        }
        total += all;
    }
    if (threadIdx.x == 0) counts[b] = total;
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

}  // namespace

extern "C" int dh_centernet_pool_nms(const float* heat, float* out, int64_t planes, int W, int H, int kernel, void* stream)
{
    if (planes < 0 || W < 1 || H < 1 || kernel < 1 || (kernel & 1) == 0 || planes > (1 << 30)) return DH_ERR_ARG;
    if (planes == 0) return DH_OK;
    if (!heat || !out) return DH_ERR_ARG;
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
    return (size_t)B * W * H * (sizeof(float) + sizeof(int)) + 512;
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
    hipLaunchKernelGGL(k_cn_best_class, dim3((unsigned)((cells + 255) / 256), B), dim3(256), 0, st, heat, C, W, H, conf, cls);
    hipLaunchKernelGGL(k_cn_emit, dim3(B), dim3(1024), 0, st, conf, cls, size, offset, W, H, (float)confidence, det, counts);
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
    if (hipMemsetAsync(t_cls, 0, sizeof(float) * B * C * cells, st) != hipSuccess) return DH_ERR_LAUNCH;
    if (hipMemsetAsync(t_size, 0, sizeof(float) * B * 2 * cells, st) != hipSuccess) return DH_ERR_LAUNCH;
    if (hipMemsetAsync(t_offset, 0, sizeof(float) * B * 2 * cells, st) != hipSuccess) return DH_ERR_LAUNCH;
    if (hipMemsetAsync(t_mask, 0, sizeof(float) * B * cells, st) != hipSuccess) return DH_ERR_LAUNCH;
    if (hipMemsetAsync(owner_ws, 0xFF, sizeof(int) * B * cells, st) != hipSuccess) return DH_ERR_LAUNCH;
    if (Lmax == 0) return DH_OK;
    if (!labels) return DH_ERR_ARG;
    const dim3 grid((Lmax + 255) / 256, B);
    hipLaunchKernelGGL(k_cn_targets_mark, grid, dim3(256), 0, st, labels, nlabels, Lmax, (float)input_w, (float)input_h, C, W, H,
                       t_cls, t_mask, owner_ws);
    hipLaunchKernelGGL(k_cn_targets_fill, grid, dim3(256), 0, st, labels, nlabels, Lmax, (float)input_w, (float)input_h, C, W, H,
                       owner_ws, t_size, t_offset);
    return dh_launch_status();
}
