// Micro-benchmark: how fast can one 4-byte column of 340-byte rows be read?  (the objectness logit of every anchor)
// hipcc --offload-arch=gfx950 -O3 -o strided_read strided_read.hip && ./strided_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int UNROLL, int NT>
__global__ void __launch_bounds__(256) k_col(const float* __restrict__ p, long rows, int no, float* __restrict__ out)
{
    float acc = 0.f;
    const long base = (long)blockIdx.x * (256 * UNROLL) + threadIdx.x;
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long r = base + u * 256;
        if (r < rows) v[u] = NT ? __builtin_nontemporal_load(p + r * no + 4) : p[r * no + 4];
        else v[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long r = base + u * 256;
        if (r < rows) out[r] = v[u] * 2.0f;
    }
    (void)acc;
}

// one wave sweeps 64 rows; lane = row, but consecutive u take the NEXT row of the same lane (lane owns UNROLL adjacent rows)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_col_adjacent(const float* __restrict__ p, long rows, int no, float* __restrict__ out)
{
    const long first = ((long)blockIdx.x * 256 + threadIdx.x) * UNROLL;
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = (first + u < rows) ? p[(first + u) * no + 4] : 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) if (first + u < rows) out[first + u] = v[u] * 2.0f;
}

// compact column for comparison
__global__ void __launch_bounds__(256) k_compact(const float* __restrict__ p, long rows, float* __restrict__ out)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) out[r] = p[r] * 2.0f;
}

int main()
{
    const long rows = 64l * 25200;
    const int no = 85;
    float *p, *out;
    CHECK(hipMalloc(&p, rows * no * sizeof(float)));
    CHECK(hipMalloc(&out, rows * sizeof(float)));
    CHECK(hipMemset(p, 0, rows * no * sizeof(float)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    // a 1 GiB fill between launches evicts the column from the 256 MB Infinity Cache: in the pipeline the maps were
    // last touched by a streaming pass over all 548 MB, so the column is not resident either
    char* flush;
    CHECK(hipMalloc(&flush, 1l << 30));
    auto timeit = [&](const char* name, auto launch) {
        float total = 0.f;
        for (int i = 0; i < 8; ++i) {
            hipMemsetAsync(flush, i, 1l << 30, 0);
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 2) total += ms;
        }
        const float ms = total / 6;
        printf("%-34s %8.1f us   (64 B sectors: %6.1f GB/s)\n", name, ms * 1e3, rows * 64.0 / (ms * 1e-3) / 1e9);
    };
#define GRID(U) dim3((unsigned)((rows + 256 * U - 1) / (256 * U)))
    timeit("lane=row unroll 1", [&] { hipLaunchKernelGGL((k_col<1, 0>), GRID(1), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane=row unroll 4", [&] { hipLaunchKernelGGL((k_col<4, 0>), GRID(4), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane=row unroll 8", [&] { hipLaunchKernelGGL((k_col<8, 0>), GRID(8), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane=row unroll 16", [&] { hipLaunchKernelGGL((k_col<16, 0>), GRID(16), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane=row unroll 4 nontemporal", [&] { hipLaunchKernelGGL((k_col<4, 1>), GRID(4), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane=row unroll 8 nontemporal", [&] { hipLaunchKernelGGL((k_col<8, 1>), GRID(8), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane owns 4 adjacent rows", [&] { hipLaunchKernelGGL((k_col_adjacent<4>), GRID(4), dim3(256), 0, 0, p, rows, no, out); });
    timeit("lane owns 8 adjacent rows", [&] { hipLaunchKernelGGL((k_col_adjacent<8>), GRID(8), dim3(256), 0, 0, p, rows, no, out); });
    timeit("compact column", [&] { hipLaunchKernelGGL(k_compact, GRID(1), dim3(256), 0, 0, p, rows, out); });
    return 0;
}
