// Box-format helpers on (n, >=4) row-major tensors.
//
// Reference: object_detection/yolov5/utils/general.py:543-560 (xyxy2xywh, xywh2xyxy),
//            :619-644 (scale_coords, clip_coords; both modify the caller's tensor in place),
//            object_detection/yolov5/models/yolo.py:155-163 (_descale_pred, in-place branch).
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

__global__ void __launch_bounds__(256) k_box_convert(const float* __restrict__ in, long in_stride, float* __restrict__ out,
                                                     long out_stride, long n, int mode)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + i * in_stride;
    const float a = p[0], b = p[1], c = p[2], d = p[3];
    float* q = out + i * out_stride;
    if (mode == 0) {          // xywh -> xyxy
        q[0] = a - c / 2; q[1] = b - d / 2; q[2] = a + c / 2; q[3] = b + d / 2;
    } else {                  // xyxy -> xywh
        q[0] = (a + c) / 2; q[1] = (b + d) / 2; q[2] = c - a; q[3] = d - b;
    }
}

__global__ void __launch_bounds__(256) k_scale_clip(float* __restrict__ boxes, long stride, long n, int do_scale,
                                                    float pad_x, float pad_y, float gain, float max_x, float max_y)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* p = boxes + i * stride;
    float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    if (do_scale) {           // general.py:628-630
        x1 = __fdiv_rn(x1 - pad_x, gain); x2 = __fdiv_rn(x2 - pad_x, gain);
        y1 = __fdiv_rn(y1 - pad_y, gain); y2 = __fdiv_rn(y2 - pad_y, gain);
    }
    // general.py:638-641 clamp_(0, limit): min(max(v, 0), limit)
    p[0] = fminf(fmaxf(x1, 0.0f), max_x);
    p[1] = fminf(fmaxf(y1, 0.0f), max_y);
    p[2] = fminf(fmaxf(x2, 0.0f), max_x);
    p[3] = fminf(fmaxf(y2, 0.0f), max_y);
}

__global__ void __launch_bounds__(256) k_descale(float* __restrict__ rows, long stride, long n, float scale, int flips,
                                                 float img_h, float img_w)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* p = rows + i * stride;
    float x = __fdiv_rn(p[0], scale), y = __fdiv_rn(p[1], scale);                 // yolo.py:158
    const float w = __fdiv_rn(p[2], scale), h = __fdiv_rn(p[3], scale);
    if (flips == 2) y = img_h - y;                                               // yolo.py:160
    else if (flips == 3) x = img_w - x;                                          // yolo.py:162
    p[0] = x; p[1] = y; p[2] = w; p[3] = h;
}

}  // namespace

extern "C" int dh_descale_pred(float* rows, int64_t row_stride, int64_t n, double scale, int flips, double img_h, double img_w,
                               void* stream)
{
    if (n < 0 || (n > 0 && !rows) || row_stride < 4) return DH_ERR_ARG;
    if (n == 0) return DH_OK;
    hipLaunchKernelGGL(k_descale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, (long)row_stride,
                       (long)n, (float)scale, flips, (float)img_h, (float)img_w);
    return dh_launch_status();
}

extern "C" int dh_box_convert(const float* in, int64_t in_row_stride, float* out, int64_t out_row_stride, int64_t n,
                              int mode, void* stream)
{
    if (n < 0 || (mode != 0 && mode != 1) || (n > 0 && (!in || !out))) return DH_ERR_ARG;
    if (n == 0) return DH_OK;
    hipLaunchKernelGGL(k_box_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (long)in_row_stride, out, (long)out_row_stride, (long)n, mode);
    return dh_launch_status();
}

extern "C" int dh_scale_clip_coords(float* boxes, int64_t row_stride, int64_t n, int do_scale, double pad_x, double pad_y,
                                    double gain, double max_x, double max_y, void* stream)
{
    if (n < 0 || (n > 0 && !boxes)) return DH_ERR_ARG;
    if (n == 0) return DH_OK;
    hipLaunchKernelGGL(k_scale_clip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                       (long)row_stride, (long)n, do_scale, (float)pad_x, (float)pad_y, (float)gain, (float)max_x,
                       (float)max_y);
    return dh_launch_status();
}
