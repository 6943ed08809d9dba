// Micro-benchmark: what a plain float4 copy / read / write reaches on this GPU, over launch shapes.
// hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, int NT>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
    }
}

template <int U, int NT>
__global__ void __launch_bounds__(256) k_copy_stride(const f4* __restrict__ in, f4* __restrict__ out, size_t n)
{
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += (size_t)gridDim.x * 256 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
        }
    }
}

// Workgroups are dealt to the 8 XCDs round-robin by their index; here XCD x streams the x-th contiguous eighth of the
// buffer (CONTIG) instead of every eighth tile of it.
template <int U, int NT>
__global__ void __launch_bounds__(256) k_copy_xcd(const f4* __restrict__ in, f4* __restrict__ out, size_t n)
{
    const unsigned per = (gridDim.x + 7) / 8;
    const unsigned blk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const size_t base = (size_t)blk * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
    }
}

// Launch shapes around the best one: workgroups of TPB threads, U vectors per lane, either a workgroup's U tiles one after
// the other (ADJ = 0: lane i holds vectors i, i + TPB, ...) or a lane's U vectors next to each other (ADJ = 1: 16*U bytes
// per lane).
template <int TPB, int U, int ADJ>
__global__ void __launch_bounds__(TPB) k_copy_shape(const f4* __restrict__ in, f4* __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (TPB * U) + (ADJ ? threadIdx.x * U : threadIdx.x);
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * (ADJ ? 1 : TPB);
        if (i < n) v[u] = __builtin_nontemporal_load(in + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * (ADJ ? 1 : TPB);
        if (i < n) __builtin_nontemporal_store(v[u], out + i);
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    const f4 z = {0.f, 1.f, 2.f, 3.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) __builtin_nontemporal_store(z, out + i);
    }
}

int main()
{
    const size_t bytes = 64ull * 25200 * 85 * 4;     // one decoded tensor of the headline configuration
    const size_t n = bytes / 16;
    f4 *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 0, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto timeit = [&](const char* name, double moved, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms / 20 * 1e3, moved / (ms / 20 * 1e-3) / 1e9);
    };
#define G(U) dim3((unsigned)((n + 256 * U - 1) / (256 * U)))
#define COPY(U, NT) timeit("copy unroll " #U " nt " #NT, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<U, NT>), G(U), dim3(256), 0, 0, in, out, n); })
    COPY(1, 0); COPY(1, 3); COPY(2, 3); COPY(4, 0); COPY(4, 1); COPY(4, 2); COPY(4, 3); COPY(8, 0); COPY(8, 3); COPY(16, 3);
#define XCD(U, NT) timeit("copy, an eighth of the buffer per XCD, unroll " #U " nt " #NT, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_xcd<U, NT>), dim3((G(U).x + 7) / 8 * 8), dim3(256), 0, 0, in, out, n); })
    XCD(1, 3); XCD(2, 3); XCD(4, 3); XCD(2, 0);
#define SHAPE(TPB, U, ADJ) timeit("copy nt 3, " #TPB " threads, unroll " #U ", adjacent " #ADJ, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_shape<TPB, U, ADJ>), dim3((unsigned)((n + TPB * U - 1) / (TPB * U))), dim3(TPB), 0, 0, in, out, n); })
    SHAPE(64, 1, 0); SHAPE(64, 2, 0); SHAPE(64, 2, 1); SHAPE(64, 4, 0); SHAPE(128, 1, 0); SHAPE(128, 2, 0); SHAPE(128, 2, 1);
    SHAPE(256, 2, 1); SHAPE(512, 1, 0); SHAPE(512, 2, 0); SHAPE(1024, 1, 0);
#define STRIDE(U, NT, GRID) timeit("grid-stride " #GRID " blocks unroll " #U " nt " #NT, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_stride<U, NT>), dim3(GRID), dim3(256), 0, 0, in, out, n); })
    STRIDE(4, 3, 2048); STRIDE(4, 3, 4096); STRIDE(8, 3, 2048); STRIDE(4, 0, 2048); STRIDE(1, 3, 4096); STRIDE(1, 3, 8192);
    timeit("hipMemcpyAsync D2D", 2.0 * bytes, [&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    timeit("write only unroll 4 (nt)", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_write<4>), G(4), dim3(256), 0, 0, out, n); });
    timeit("hipMemsetAsync", 1.0 * bytes, [&] { hipMemsetAsync(out, 0, bytes, 0); });
    return 0;
}
