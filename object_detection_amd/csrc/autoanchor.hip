// AutoAnchor's anchor-to-label fit metric.
//
// Reference (object_detection/yolov5/utils/autoanchor.py):
//   :86-90  metric(k, wh)      r = wh[:,None]/k[None]; x = min(r, 1/r).min(2); best = x.max(1)
//   :92-94  anchor_fitness     mean(best * (best > thr))
//   :35-41  check metric       bpr = mean(best > thr), aat = mean(sum_anchors(x > thr))
// One thread per label, the anchors in LDS; x and best are written only when asked for, the three sums always
// (float64 accumulators: the sum of best*(best>thr), the number of labels with best > thr, the number of x > thr).
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

#define AA_MAX_ANCHORS 64

__global__ void __launch_bounds__(256) k_anchor_metric(const float* __restrict__ wh, long N, const float* __restrict__ k, int n,
                                                       float thr, float* __restrict__ x_out, float* __restrict__ best_out,
                                                       double* __restrict__ sums)
{
    __shared__ float s_k[AA_MAX_ANCHORS * 2];
    __shared__ double s_red[3][4];
    if (threadIdx.x < n * 2) s_k[threadIdx.x] = k[threadIdx.x];
    __syncthreads();
    const long i = blockIdx.x * 256l + threadIdx.x;
    double fit = 0.0, n_best = 0.0, n_x = 0.0;
    if (i < N) {
        const float w = wh[i * 2], h = wh[i * 2 + 1];
        float best = 0.0f;
        int above = 0;
        for (int a = 0; a < n; ++a) {
            const float r0 = __fdiv_rn(w, s_k[a * 2]), r1 = __fdiv_rn(h, s_k[a * 2 + 1]);
            const float m = fminf(fminf(r0, __fdiv_rn(1.0f, r0)), fminf(r1, __fdiv_rn(1.0f, r1)));
            if (x_out) x_out[i * n + a] = m;
            best = a == 0 ? m : fmaxf(best, m);
            above += m > thr;
        }
        if (best_out) best_out[i] = best;
        if (best > thr) { fit = (double)best; n_best = 1.0; }
        n_x = (double)above;
    }
    fit = dh_wave_sum_d(fit);
    n_best = dh_wave_sum_d(n_best);
    n_x = dh_wave_sum_d(n_x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wave] = fit; s_red[1][wave] = n_best; s_red[2][wave] = n_x; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = (s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + (s_red[threadIdx.x][2] + s_red[threadIdx.x][3]);
        if (v != 0.0) atomicAdd(&sums[threadIdx.x], v);
    }
}

}  // namespace

extern "C" int dh_anchor_metric(const float* wh, int64_t N, const float* k, int n, double thr, float* x, float* best,
                                double* sums, void* stream)
{
    if (N < 0 || n < 1 || n > AA_MAX_ANCHORS || !k || !sums || (N > 0 && !wh) || N * (int64_t)n >= (1ll << 40)) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dh_fill_async(sums, 0, sizeof(double) * 3, st) != DH_OK) return DH_ERR_LAUNCH;
    if (N == 0) return DH_OK;
    hipLaunchKernelGGL(k_anchor_metric, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, wh, (long)N, k, n, (float)thr, x, best,
                       sums);
    return dh_launch_status();
}
