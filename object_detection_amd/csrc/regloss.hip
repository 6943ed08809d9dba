// Masked L1 / smooth-L1 regression loss of the CenterNet size and offset heads.
//
// Reference (object_detection/loss.py:86-100, RegressionLossWithMask.forward):
//   pred, target (B, 2, W, H) -> (B, W, H, 2);  m = mask (B, W, H) repeated over the two channels
//   loss = sum(l(pred*m, target*m)) / (sum(m) * 2 + 1e-4),  l = |d| or smooth-L1 with beta 1 (0.5 d^2 if |d| < 1 else |d| - 0.5)
// One pass over the maps (16 B per cell of pred and target, 4 B of mask); the backward is one more pass:
//   d loss / d pred = m * l'(d) / denominator,  l'(d) = sign(d) (0 at 0), or d inside (-1, 1) for the smooth form.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

struct rl_args {
    const float* pred;      // (B, 2, S)
    const float* target;
    const float* mask;      // (B, S)
    long cells;             // B * S
    long S;
    int smooth;
};

__global__ void __launch_bounds__(256) k_regloss_fwd(const rl_args a, double* __restrict__ partial)
{
    const long cell = blockIdx.x * 256l + threadIdx.x;
    float loss = 0.0f, msum = 0.0f;
    if (cell < a.cells) {
        const long b = cell / a.S, s = cell - b * a.S;
        const float m = a.mask[cell];
        msum = m;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const size_t i = ((size_t)b * 2 + c) * a.S + s;
            const float d = fabsf(a.pred[i] * m - a.target[i] * m);                    // loss.py:92-98
            loss += a.smooth ? (d < 1.0f ? 0.5f * d * d : d - 0.5f) : d;
        }
    }
    double l = dh_wave_sum_d((double)loss), ms = dh_wave_sum_d((double)msum);
    __shared__ double s_p[2][4];
    if ((threadIdx.x & 63) == 0) { s_p[0][threadIdx.x >> 6] = l; s_p[1][threadIdx.x >> 6] = ms; }
    __syncthreads();
    if (threadIdx.x < 2) partial[(size_t)blockIdx.x * 2 + threadIdx.x] =
        (s_p[threadIdx.x][0] + s_p[threadIdx.x][1]) + (s_p[threadIdx.x][2] + s_p[threadIdx.x][3]);
}

// out2: [0] the loss, [1] the denominator (kept for the backward)
__global__ void __launch_bounds__(1024) k_regloss_final(const double* __restrict__ partial, long nblocks, float* __restrict__ out2)
{
    __shared__ double s_p[2][16];
    double l = 0.0, ms = 0.0;
    for (long i = threadIdx.x; i < nblocks; i += 1024) { l += partial[i * 2]; ms += partial[i * 2 + 1]; }
    l = dh_wave_sum_d(l);
    ms = dh_wave_sum_d(ms);
    if ((threadIdx.x & 63) == 0) { s_p[0][threadIdx.x >> 6] = l; s_p[1][threadIdx.x >> 6] = ms; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tl = 0.0, tm = 0.0;
        for (int w = 0; w < 16; ++w) { tl += s_p[0][w]; tm += s_p[1][w]; }
        const float denom = (float)(tm * 2.0) + 1e-4f;                                  // loss.py:99: expand_mask.sum() + 1e-4
        out2[0] = (float)tl / denom;
        out2[1] = denom;
    }
}

__global__ void __launch_bounds__(256) k_regloss_bwd(const rl_args a, const float* __restrict__ out2, const float* __restrict__ grad_out,
                                                     float* __restrict__ grad)
{
    const long cell = blockIdx.x * 256l + threadIdx.x;
    if (cell >= a.cells) return;
    const long b = cell / a.S, s = cell - b * a.S;
    const float m = a.mask[cell];
    const float scale = grad_out[0] / out2[1];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const size_t i = ((size_t)b * 2 + c) * a.S + s;
        const float d = a.pred[i] * m - a.target[i] * m;
        float g = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        if (a.smooth && fabsf(d) < 1.0f) g = d;
        grad[i] = scale * g * m;
    }
}

}  // namespace

extern "C" size_t dh_regression_loss_workspace(int64_t cells)
{
    if (cells <= 0) return 0;
    return (size_t)((cells + 255) / 256) * 2 * sizeof(double) + 256;
}

extern "C" int dh_regression_loss_fwd(const float* pred, const float* target, const float* mask, int64_t B, int64_t S, int smooth,
                                      float* out2, void* workspace, size_t workspace_bytes, void* stream)
{
    if (B < 0 || S < 0 || B * S >= (1ll << 40) || !out2) return DH_ERR_ARG;
    const long cells = (long)(B * S);
    hipStream_t st = (hipStream_t)stream;
    const long blocks = (cells + 255) / 256;
    if (cells > 0 && (!pred || !target || !mask)) return DH_ERR_ARG;
    if (cells > 0 && (!workspace || workspace_bytes < dh_regression_loss_workspace(cells))) return DH_ERR_WORKSPACE;
    rl_args a = {pred, target, mask, cells, (long)S, smooth};
    if (cells > 0) hipLaunchKernelGGL(k_regloss_fwd, dim3((unsigned)blocks), dim3(256), 0, st, a, (double*)workspace);
    hipLaunchKernelGGL(k_regloss_final, dim3(1), dim3(1024), 0, st, (const double*)workspace, blocks, out2);
    return dh_launch_status();
}

extern "C" int dh_regression_loss_bwd(const float* pred, const float* target, const float* mask, int64_t B, int64_t S, int smooth,
                                      const float* out2, const float* grad_out, float* grad_pred, void* stream)
{
    if (B < 0 || S < 0 || B * S >= (1ll << 40)) return DH_ERR_ARG;
    const long cells = (long)(B * S);
    if (cells == 0) return DH_OK;
    if (!pred || !target || !mask || !out2 || !grad_out || !grad_pred) return DH_ERR_ARG;
    rl_args a = {pred, target, mask, cells, (long)S, smooth};
    hipLaunchKernelGGL(k_regloss_bwd, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out2, grad_out,
                       grad_pred);
    return dh_launch_status();
}
