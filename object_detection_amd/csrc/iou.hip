// Element-wise IoU-family kernels (value + gradient) and the fused CenterNet masked-CIoU loss.
//
// Reference behaviour:
//   object_detection/yolov5/utils/metrics.py:201-240   bbox_iou on (4,n) vs (n,4)
//   object_detection/loss.py:118-166                   bbox_iou on [...,4]
//   object_detection/loss.py:102-115                   IoULossWithMask.forward
//   object_detection/detectors/centernet/loss.py:4-28  offset_to_coord + compute_losses (box branch)
// All kernels are HBM-streaming: every box component is read once with unit-stride (SoA planes)
// or 16-byte-per-lane (AoS rows) accesses; nothing is re-read.
#include "dh_iou.h"
#include "../../include/dethead.h"

namespace {

struct box_layout {
    long comp;   // stride between components of one box, in floats
    long elem;   // stride between boxes, in floats
};

__device__ __forceinline__ void load_box(const float* __restrict__ p, const box_layout l, long i, float v[4])
{
    if (l.comp == 1 && l.elem == 4) {   // AoS rows: one 16-byte load per lane
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const float* q = p + i * l.elem;
        v[0] = q[0]; v[1] = q[l.comp]; v[2] = q[2 * l.comp]; v[3] = q[3 * l.comp];
    }
}

__device__ __forceinline__ dh_corners as_corners(const float v[4], int xyxy)
{
    if (xyxy) { dh_corners c = {v[0], v[1], v[2], v[3]}; return c; }
    return dh_from_xywh(v[0], v[1], v[2], v[3]);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_bbox_iou_fwd(const float* __restrict__ b1, box_layout l1,
                                                       const float* __restrict__ b2, box_layout l2,
                                                       float* __restrict__ out, long n, int xyxy, float eps, float ope)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float p[4], t[4];
        load_box(b1, l1, i, p);
        load_box(b2, l2, i, t);
        out[i] = dh_iou_value<MODE>(as_corners(p, xyxy), as_corners(t, xyxy), eps, ope);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_bbox_iou_bwd(const float* __restrict__ b1, box_layout l1,
                                                       const float* __restrict__ b2, box_layout l2,
                                                       const float* __restrict__ gout, float* __restrict__ g1,
                                                       box_layout lg, long n, int xyxy, float eps, float ope)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float p[4], t[4];
        load_box(b1, l1, i, p);
        load_box(b2, l2, i, t);
        dh_corners gc;
        dh_iou_value_grad<MODE>(as_corners(p, xyxy), as_corners(t, xyxy), eps, ope, gout[i], gc);
        float r[4];
        if (xyxy) { r[0] = gc.x1; r[1] = gc.y1; r[2] = gc.x2; r[3] = gc.y2; }
        else dh_grad_to_xywh(gc, r[0], r[1], r[2], r[3]);
        if (lg.comp == 1 && lg.elem == 4) {
            *reinterpret_cast<float4*>(g1 + 4 * i) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            float* q = g1 + i * lg.elem;
            q[0] = r[0]; q[lg.comp] = r[1]; q[2 * lg.comp] = r[2]; q[3 * lg.comp] = r[3];
        }
    }
}

int grid_for(long n, int block, int per_thread)
{
    long g = (n + (long)block * per_thread - 1) / ((long)block * per_thread);
    if (g < 1) g = 1;
    // no persistent-grid cap: on this chip one element per thread and as many workgroups as it takes streams faster
    // than a capped grid that loops (a plain copy reaches 6.1-6.7 TB/s this way against 5.3-6.0 grid-stride)
    if (g > 0x7fffffffL) g = 0x7fffffffL;
    return (int)g;
}

bool vec_ok(const float* p, long comp, long elem) { return !(comp == 1 && elem == 4) || dh_aligned16(p); }

// ---------------------------------------------------------------- CenterNet masked CIoU loss
// Maps are (B, 2, W, H) planes (x-plane then y-plane); one thread owns 4 consecutive cells of a plane.
struct cn_maps {
    const float* pxy; long pxy_b;   // prediction centre planes (offset or absolute), batch stride in floats
    const float* pwh; long pwh_b;
    const float* txy; long txy_b;
    const float* twh; long twh_b;
    const float* mask;              // (B, W, H)
};

struct cn_cell {
    dh_corners a, b;
    float m;
};

__device__ __forceinline__ float4 cn_ld4(const float* p)
{
    const dh_f4 v = __builtin_nontemporal_load(reinterpret_cast<const dh_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void cn_load4(const cn_maps& mp, int b, long cell, long plane, dh_fastdiv H, int add_coords,
                                         float px[4], float py[4], float pw[4], float ph[4], float tx[4], float ty[4],
                                         float tw[4], float th[4], float m[4])
{
    const float4 a0 = cn_ld4(mp.pxy + b * mp.pxy_b + cell);
    const float4 a1 = cn_ld4(mp.pxy + b * mp.pxy_b + plane + cell);
    const float4 a2 = cn_ld4(mp.pwh + b * mp.pwh_b + cell);
    const float4 a3 = cn_ld4(mp.pwh + b * mp.pwh_b + plane + cell);
    const float4 c0 = cn_ld4(mp.txy + b * mp.txy_b + cell);
    const float4 c1 = cn_ld4(mp.txy + b * mp.txy_b + plane + cell);
    const float4 c2 = cn_ld4(mp.twh + b * mp.twh_b + cell);
    const float4 c3 = cn_ld4(mp.twh + b * mp.twh_b + plane + cell);
    const float4 mm = cn_ld4(mp.mask + b * plane + cell);
    px[0] = a0.x; px[1] = a0.y; px[2] = a0.z; px[3] = a0.w;
    py[0] = a1.x; py[1] = a1.y; py[2] = a1.z; py[3] = a1.w;
    pw[0] = a2.x; pw[1] = a2.y; pw[2] = a2.z; pw[3] = a2.w;
    ph[0] = a3.x; ph[1] = a3.y; ph[2] = a3.z; ph[3] = a3.w;
    tx[0] = c0.x; tx[1] = c0.y; tx[2] = c0.z; tx[3] = c0.w;
    ty[0] = c1.x; ty[1] = c1.y; ty[2] = c1.z; ty[3] = c1.w;
    tw[0] = c2.x; tw[1] = c2.y; tw[2] = c2.z; tw[3] = c2.w;
    th[0] = c3.x; th[1] = c3.y; th[2] = c3.z; th[3] = c3.w;
    m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
    if (add_coords) {   // centernet/loss.py:4-13: plane 0 += index along dim 2, plane 1 += index along dim 3
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = (uint32_t)cell + k;
            const uint32_t qx = dh_div(c, H);
            const float ix = (float)qx, iy = (float)(c - qx * H.d);
            px[k] += ix; py[k] += iy; tx[k] += ix; ty[k] += iy;
        }
    }
}

// A cell whose two (masked) boxes are exactly zero contributes exactly 0 to the CIoU sum
// (inter 0, union eps, rho2 0, both atan 0), so the sparse-mask common case skips the arithmetic.
__device__ __forceinline__ bool cn_all_zero(float a, float b, float c, float d, float e, float f, float g, float h)
{
    return a == 0.0f && b == 0.0f && c == 0.0f && d == 0.0f && e == 0.0f && f == 0.0f && g == 0.0f && h == 0.0f;
}

__global__ void __launch_bounds__(256) k_cn_loss_fwd(cn_maps mp, int W, dh_fastdiv Hd, int add_coords, float eps,
                                                     float ope, double* __restrict__ partial)
{
    const long plane = (long)W * Hd.d;
    const long vec_per_img = plane / 4;
    const int b = blockIdx.y;
    double s_iou = 0.0, s_mask = 0.0;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < vec_per_img; v += (long)gridDim.x * blockDim.x) {
        const long cell = v * 4;
        float px[4], py[4], pw[4], ph[4], tx[4], ty[4], tw[4], th[4], m[4];
        cn_load4(mp, b, cell, plane, Hd, add_coords, px, py, pw, ph, tx, ty, tw, th, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ax = px[k] * m[k], ay = py[k] * m[k], aw = pw[k] * m[k], ah = ph[k] * m[k];
            const float bx = tx[k] * m[k], by = ty[k] * m[k], bw = tw[k] * m[k], bh = th[k] * m[k];
            s_mask += (double)m[k];
            if (cn_all_zero(ax, ay, aw, ah, bx, by, bw, bh)) continue;
            s_iou += (double)dh_iou_value<DH_CIOU>(dh_from_xywh(ax, ay, aw, ah), dh_from_xywh(bx, by, bw, bh), eps, ope);
        }
    }
    __shared__ double red[2][4];
    s_iou = dh_wave_sum_d(s_iou);
    s_mask = dh_wave_sum_d(s_mask);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s_iou; red[1][w] = s_mask; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long slot = (long)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * slot + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[2 * slot + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// The reduction of the forward's per-workgroup partial sums by the 256 threads of one workgroup (all of them call it):
// returns the divisor 4*sum_mask + 1e-4 of object_detection/loss.py:112-115 to every thread and, when `out` is given, thread
// 0 writes out[0] = 1 - sum_ciou / divisor and out[1] = divisor.  One order of additions wherever it runs, so the
// reduction launch and the workgroups of the fused backward arrive at the same bits.
__device__ __forceinline__ float cn_reduce(const double* __restrict__ partial, int nblocks, float* __restrict__ out, double (*red)[4],
                                           float* __restrict__ shared_denom)
{
    double s = 0.0, m = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        if (out) s += partial[2 * i];
        m += partial[2 * i + 1];
    }
    s = dh_wave_sum_d(s);
    m = dh_wave_sum_d(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s; red[1][w] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sum_iou = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        const float sum_m4 = (float)(4.0 * (red[1][0] + red[1][1] + red[1][2] + red[1][3]));
        const float denom = sum_m4 + 1e-4f;
        if (out) { out[0] = 1.0f - __fdiv_rn(sum_iou, denom); out[1] = denom; }
        *shared_denom = denom;
    }
    __syncthreads();
    return *shared_denom;
}

// out[0] = 1 - sum_ciou / (4*sum_mask + 1e-4)  (object_detection/loss.py:112-115); out[1] = that divisor
__global__ void __launch_bounds__(256) k_cn_loss_final(const double* __restrict__ partial, int nblocks, float* __restrict__ out)
{
    __shared__ double red[2][4];
    __shared__ float denom;
    cn_reduce(partial, nblocks, out, red, &denom);
}

struct cn_grads {
    float* gxy; long gxy_b;
    float* gwh; long gwh_b;
};

// d loss / d pred = -(g_out / denom) * dCIoU/d(pred*mask) * mask
// FUSED: the forward's partial sums have not been reduced yet.  Every workgroup adds up the mask partials itself (a few KB
// from L2, the same additions in the same order everywhere) instead of waiting for a reduction launch, and workgroup (0,0)
// also writes the loss value and the divisor to fwd_out.  g_out may then be null: upstream gradient 1.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_cn_loss_bwd(cn_maps mp, cn_grads gr, int W, dh_fastdiv Hd, int add_coords,
                                                     float eps, float ope, float* __restrict__ fwd_out,
                                                     const float* __restrict__ g_out, const double* __restrict__ partial, int nblocks)
{
    const long plane = (long)W * Hd.d;
    const long vec_per_img = plane / 4;
    const int b = blockIdx.y;
    float denom;
    if (FUSED) {
        __shared__ double red[2][4];
        __shared__ float sh_denom;
        denom = cn_reduce(partial, nblocks, (blockIdx.x == 0 && blockIdx.y == 0) ? fwd_out : nullptr, red, &sh_denom);
    } else {
        denom = fwd_out[1];
    }
    const float scale = -__fdiv_rn(g_out ? g_out[0] : 1.0f, denom);
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < vec_per_img; v += (long)gridDim.x * blockDim.x) {
        const long cell = v * 4;
        float px[4], py[4], pw[4], ph[4], tx[4], ty[4], tw[4], th[4], m[4];
        cn_load4(mp, b, cell, plane, Hd, add_coords, px, py, pw, ph, tx, ty, tw, th, m);
        float ox[4], oy[4], ow[4], oh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ax = px[k] * m[k], ay = py[k] * m[k], aw = pw[k] * m[k], ah = ph[k] * m[k];
            const float bx = tx[k] * m[k], by = ty[k] * m[k], bw = tw[k] * m[k], bh = th[k] * m[k];
            if (m[k] == 0.0f && cn_all_zero(ax, ay, aw, ah, bx, by, bw, bh)) {   // finite gradient times a zero mask
                ox[k] = oy[k] = ow[k] = oh[k] = 0.0f;
                continue;
            }
            dh_corners gc;
            dh_iou_value_grad<DH_CIOU>(dh_from_xywh(ax, ay, aw, ah), dh_from_xywh(bx, by, bw, bh), eps, ope, scale, gc);
            float gx, gy, gw, gh;
            dh_grad_to_xywh(gc, gx, gy, gw, gh);
            ox[k] = gx * m[k]; oy[k] = gy * m[k]; ow[k] = gw * m[k]; oh[k] = gh * m[k];
        }
        *reinterpret_cast<float4*>(gr.gxy + b * gr.gxy_b + cell) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4*>(gr.gxy + b * gr.gxy_b + plane + cell) = make_float4(oy[0], oy[1], oy[2], oy[3]);
        *reinterpret_cast<float4*>(gr.gwh + b * gr.gwh_b + cell) = make_float4(ow[0], ow[1], ow[2], ow[3]);
        *reinterpret_cast<float4*>(gr.gwh + b * gr.gwh_b + plane + cell) = make_float4(oh[0], oh[1], oh[2], oh[3]);
    }
}

// blocks along x per image: enough to fill the chip at the batch size given, at most 8 per CU overall
int cn_grid_x(int B, int W, int H)
{
    const long vecs = (long)W * H / 4;
    long gx = (vecs + 255) / 256;
    const long cap = (2048 + B - 1) / B;
    if (gx > cap) gx = cap;
    return (int)(gx < 1 ? 1 : gx);
}

bool cn_args_ok(const cn_maps& mp, int B, int W, int H)
{
    if (B <= 0 || B > 65535 || W <= 0 || H <= 0) return false;
    if (((long)W * H) % 4 != 0 || (long)W * H >= (1L << 31)) return false;
    const long strides[4] = {mp.pxy_b, mp.pwh_b, mp.txy_b, mp.twh_b};
    for (long s : strides) if (s % 4 != 0 || s < 2L * W * H) return false;
    return mp.pxy && mp.pwh && mp.txy && mp.twh && mp.mask;
}

bool cn_aligned(const cn_maps& mp)
{
    return dh_aligned16(mp.pxy) && dh_aligned16(mp.pwh) && dh_aligned16(mp.txy) && dh_aligned16(mp.twh) && dh_aligned16(mp.mask);
}

}  // namespace

#define DH_DISPATCH_MODE(mode, CALL)            \
    switch (mode) {                             \
    case DH_IOU: CALL(DH_IOU); break;           \
    case DH_GIOU: CALL(DH_GIOU); break;         \
    case DH_DIOU: CALL(DH_DIOU); break;         \
    case DH_CIOU: CALL(DH_CIOU); break;         \
    default: return DH_ERR_ARG;                 \
    }

extern "C" int dh_bbox_iou_fwd(const float* box1, int64_t s1_comp, int64_t s1_elem, const float* box2, int64_t s2_comp,
                               int64_t s2_elem, float* out, int64_t n, int xyxy, int mode, double eps, void* stream)
{
    if (n < 0 || (n > 0 && (!box1 || !box2 || !out))) return DH_ERR_ARG;
    if (n == 0) return DH_OK;
    if (!vec_ok(box1, s1_comp, s1_elem) || !vec_ok(box2, s2_comp, s2_elem)) return DH_ERR_ALIGN;
    const box_layout l1 = {s1_comp, s1_elem}, l2 = {s2_comp, s2_elem};
    const float e = (float)eps, ope = (float)(1.0 + eps);
    hipStream_t st = (hipStream_t)stream;
#define CALL(M) hipLaunchKernelGGL(k_bbox_iou_fwd<M>, dim3(grid_for(n, 256, 1)), dim3(256), 0, st, box1, l1, box2, l2, out, (long)n, xyxy, e, ope)
    DH_DISPATCH_MODE(mode, CALL)
#undef CALL
    return dh_launch_status();
}

extern "C" int dh_bbox_iou_bwd(const float* box1, int64_t s1_comp, int64_t s1_elem, const float* box2, int64_t s2_comp,
                               int64_t s2_elem, const float* grad_out, float* grad_box1, int64_t sg_comp,
                               int64_t sg_elem, int64_t n, int xyxy, int mode, double eps, void* stream)
{
    if (n < 0 || (n > 0 && (!box1 || !box2 || !grad_out || !grad_box1))) return DH_ERR_ARG;
    if (n == 0) return DH_OK;
    if (!vec_ok(box1, s1_comp, s1_elem) || !vec_ok(box2, s2_comp, s2_elem) || !vec_ok(grad_box1, sg_comp, sg_elem))
        return DH_ERR_ALIGN;
    const box_layout l1 = {s1_comp, s1_elem}, l2 = {s2_comp, s2_elem}, lg = {sg_comp, sg_elem};
    const float e = (float)eps, ope = (float)(1.0 + eps);
    hipStream_t st = (hipStream_t)stream;
#define CALL(M) hipLaunchKernelGGL(k_bbox_iou_bwd<M>, dim3(grid_for(n, 256, 1)), dim3(256), 0, st, box1, l1, box2, l2, grad_out, grad_box1, lg, (long)n, xyxy, e, ope)
    DH_DISPATCH_MODE(mode, CALL)
#undef CALL
    return dh_launch_status();
}

extern "C" size_t dh_centernet_box_loss_workspace(int B, int W, int H)
{
    if (B <= 0 || W <= 0 || H <= 0) return 0;
    return (size_t)cn_grid_x(B, W, H) * B * 2 * sizeof(double);
}

extern "C" int dh_centernet_box_loss_fwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                                         int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                                         const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W,
                                         int H, int add_coords, double eps, float* out2, void* workspace,
                                         size_t workspace_bytes, void* stream)
{
    const cn_maps mp = {pred_xy, pred_xy_bstride, pred_wh, pred_wh_bstride, tgt_xy, tgt_xy_bstride, tgt_wh, tgt_wh_bstride, mask};
    if (!cn_args_ok(mp, B, W, H) || !out2 || !workspace) return DH_ERR_ARG;
    if (!cn_aligned(mp) || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return DH_ERR_ALIGN;
    if (workspace_bytes < dh_centernet_box_loss_workspace(B, W, H)) return DH_ERR_WORKSPACE;
    const int gx = cn_grid_x(B, W, H);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cn_loss_fwd, dim3(gx, B), dim3(256), 0, st, mp, W, dh_make_fastdiv((uint32_t)H), add_coords,
                       (float)eps, (float)(1.0 + eps), (double*)workspace);
    hipLaunchKernelGGL(k_cn_loss_final, dim3(1), dim3(256), 0, st, (const double*)workspace, gx * B, out2);
    return dh_launch_status();
}

extern "C" int dh_centernet_box_loss_bwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                                         int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                                         const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W,
                                         int H, int add_coords, double eps, const float* fwd_out2,
                                         const float* grad_out, float* grad_xy, int64_t grad_xy_bstride,
                                         float* grad_wh, int64_t grad_wh_bstride, void* stream)
{
    const cn_maps mp = {pred_xy, pred_xy_bstride, pred_wh, pred_wh_bstride, tgt_xy, tgt_xy_bstride, tgt_wh, tgt_wh_bstride, mask};
    if (!cn_args_ok(mp, B, W, H) || !fwd_out2 || !grad_out || !grad_xy || !grad_wh) return DH_ERR_ARG;
    if (grad_xy_bstride % 4 || grad_wh_bstride % 4 || grad_xy_bstride < 2ll * W * H || grad_wh_bstride < 2ll * W * H) return DH_ERR_ARG;
    if (!cn_aligned(mp) || !dh_aligned16(grad_xy) || !dh_aligned16(grad_wh)) return DH_ERR_ALIGN;
    const cn_grads gr = {grad_xy, grad_xy_bstride, grad_wh, grad_wh_bstride};
    hipLaunchKernelGGL(k_cn_loss_bwd<false>, dim3(cn_grid_x(B, W, H), B), dim3(256), 0, (hipStream_t)stream, mp, gr, W,
                       dh_make_fastdiv((uint32_t)H), add_coords, (float)eps, (float)(1.0 + eps), const_cast<float*>(fwd_out2), grad_out,
                       (const double*)nullptr, 0);
    return dh_launch_status();
}

// Forward, reduction and gradient in two launches: the forward's partial sums, then the gradient pass, whose workgroups
// reduce the partials themselves (no reduction launch, no second stream round trip).  Same values as the two entries above.
extern "C" int dh_centernet_box_loss_fwd_bwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                                             int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                                             const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W,
                                             int H, int add_coords, double eps, const float* grad_out, float* out2,
                                             float* grad_xy, int64_t grad_xy_bstride, float* grad_wh, int64_t grad_wh_bstride,
                                             void* workspace, size_t workspace_bytes, void* stream)
{
    const cn_maps mp = {pred_xy, pred_xy_bstride, pred_wh, pred_wh_bstride, tgt_xy, tgt_xy_bstride, tgt_wh, tgt_wh_bstride, mask};
    if (!cn_args_ok(mp, B, W, H) || !out2 || !workspace || !grad_xy || !grad_wh) return DH_ERR_ARG;
    if (grad_xy_bstride % 4 || grad_wh_bstride % 4 || grad_xy_bstride < 2ll * W * H || grad_wh_bstride < 2ll * W * H) return DH_ERR_ARG;
    if (!cn_aligned(mp) || !dh_aligned16(grad_xy) || !dh_aligned16(grad_wh) || (reinterpret_cast<uintptr_t>(workspace) & 7u))
        return DH_ERR_ALIGN;
    if (workspace_bytes < dh_centernet_box_loss_workspace(B, W, H)) return DH_ERR_WORKSPACE;
    const int gx = cn_grid_x(B, W, H);
    const dh_fastdiv Hd = dh_make_fastdiv((uint32_t)H);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cn_loss_fwd, dim3(gx, B), dim3(256), 0, st, mp, W, Hd, add_coords, (float)eps, (float)(1.0 + eps),
                       (double*)workspace);
    const cn_grads gr = {grad_xy, grad_xy_bstride, grad_wh, grad_wh_bstride};
    hipLaunchKernelGGL(k_cn_loss_bwd<true>, dim3(gx, B), dim3(256), 0, st, mp, gr, W, Hd, add_coords, (float)eps, (float)(1.0 + eps),
                       out2, grad_out, (const double*)workspace, gx * B);
    return dh_launch_status();
}
