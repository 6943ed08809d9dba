// N x M pairwise IoU matrices with LDS-staged row tiles.
//
// Reference: object_detection/yolov5/utils/metrics.py:242-264 (box_iou: xyxy, no eps)
//            object_detection/yolov5/utils/metrics.py:291-296 (wh_iou: width/height only)
// The kernel is write-bound (4*N*M bytes out against 16*(N+M) in): a workgroup owns a tile of
// 32 rows x 256 columns, stages its 32 row boxes (+areas) in LDS once, each thread keeps one column
// box in registers and walks the rows, so every store instruction writes 256 consecutive floats.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

#define PW_ROWS 32
#define PW_COLS 256

template <bool WH_ONLY>
__global__ void __launch_bounds__(PW_COLS) k_pairwise_iou(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int N, int M)
{
    __shared__ float4 rows[PW_ROWS];
    __shared__ float rarea[PW_ROWS];
    const int row0 = blockIdx.y * PW_ROWS;
    const int col = blockIdx.x * PW_COLS + threadIdx.x;
    if (threadIdx.x < PW_ROWS) {
        const int r = row0 + threadIdx.x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) {
            if (WH_ONLY) { v.x = a[2 * (size_t)r]; v.y = a[2 * (size_t)r + 1]; }
            else { const float* p = a + 4 * (size_t)r; v = make_float4(p[0], p[1], p[2], p[3]); }
        }
        rows[threadIdx.x] = v;
        rarea[threadIdx.x] = WH_ONLY ? v.x * v.y : (v.z - v.x) * (v.w - v.y);
    }
    __syncthreads();
    if (col >= M) return;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WH_ONLY) { c.x = b[2 * (size_t)col]; c.y = b[2 * (size_t)col + 1]; }
    else { const float* p = b + 4 * (size_t)col; c = make_float4(p[0], p[1], p[2], p[3]); }
    const float carea = WH_ONLY ? c.x * c.y : (c.z - c.x) * (c.w - c.y);
    const int nrows = (N - row0) < PW_ROWS ? (N - row0) : PW_ROWS;
#pragma unroll 4
    for (int k = 0; k < nrows; ++k) {
        const float4 r = rows[k];
        float inter;
        if (WH_ONLY) inter = fminf(r.x, c.x) * fminf(r.y, c.y);
        else inter = fmaxf(fminf(r.z, c.z) - fmaxf(r.x, c.x), 0.0f) * fmaxf(fminf(r.w, c.w) - fmaxf(r.y, c.y), 0.0f);
        out[(size_t)(row0 + k) * M + col] = __fdiv_rn(inter, rarea[k] + carea - inter);
    }
}

int launch(bool wh, const float* a, const float* b, float* out, int64_t N, int64_t M, void* stream)
{
    if (N < 0 || M < 0 || N >= (1ll << 31) || M >= (1ll << 31)) return DH_ERR_ARG;
    if (N == 0 || M == 0) return DH_OK;
    if (!a || !b || !out) return DH_ERR_ARG;
    const long gy = (N + PW_ROWS - 1) / PW_ROWS;
    if (gy > 65535) return DH_ERR_ARG;
    const dim3 grid((unsigned)((M + PW_COLS - 1) / PW_COLS), (unsigned)gy);
    if (wh) hipLaunchKernelGGL(k_pairwise_iou<true>, grid, dim3(PW_COLS), 0, (hipStream_t)stream, a, b, out, (int)N, (int)M);
    else hipLaunchKernelGGL(k_pairwise_iou<false>, grid, dim3(PW_COLS), 0, (hipStream_t)stream, a, b, out, (int)N, (int)M);
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
