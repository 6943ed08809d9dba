// Read bandwidth of one large buffer under the access patterns a streaming kernel can choose from: how much each
// workgroup reads in one piece, how many loads a lane keeps in flight, the cache hint, grid-stride or chunked order.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// each workgroup: TILES tiles of (THREADS * U) vectors; STRIDED: tile t of workgroup b is tile (t * gridDim.x + b) of the buffer
template <int THREADS, int U, int TILES, bool NT, bool STRIDED, bool PREFETCH>
__global__ void __launch_bounds__(THREADS) k_read(const f4* __restrict__ src, size_t nvec, float* __restrict__ sink)
{
    float acc = 0.0f;
    f4 a[U], b2[U];
    auto issue = [&](int tile, f4 (&x)[U]) {
        const size_t t = STRIDED ? (size_t)tile * gridDim.x + blockIdx.x : (size_t)blockIdx.x * TILES + tile;
        const size_t v0 = t * (THREADS * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t v = v0 + (size_t)u * THREADS;
            x[u] = ld<NT>(src + (v < nvec ? v : 0));
        }
    };
    auto use = [&](const f4 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += fmaxf(fmaxf(x[u].x, x[u].y), fmaxf(x[u].z, x[u].w));
    };
    if (PREFETCH) {
        issue(0, a);
#pragma unroll 1
        for (int tile = 0; tile < TILES; tile += 2) {
            if (tile + 1 < TILES) issue(tile + 1, b2);
            use(a);
            if (tile + 1 >= TILES) break;
            if (tile + 2 < TILES) issue(tile + 2, a);
            use(b2);
        }
    } else {
#pragma unroll 1
        for (int tile = 0; tile < TILES; ++tile) { issue(tile, a); use(a); }
    }
    if (acc == 123456.789f) sink[0] = acc;
}

template <int THREADS, int U, int TILES, bool NT, bool STRIDED, bool PREFETCH>
void run(const char* name, const f4* src, size_t nvec, float* sink)
{
    const size_t per_wg = (size_t)THREADS * U * TILES;
    const unsigned grid = (unsigned)((nvec + per_wg - 1) / per_wg);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<THREADS, U, TILES, NT, STRIDED, PREFETCH>), dim3(grid), dim3(THREADS), 0, 0, src, nvec, sink);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<THREADS, U, TILES, NT, STRIDED, PREFETCH>), dim3(grid), dim3(THREADS), 0, 0, src, nvec, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-58s grid %7u  %8.1f us  %7.2f TB/s\n", name, grid, us, nvec * 16.0 / us * 1e-6);
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atol(argv[1]) : 523;
    const size_t nvec = mib * 1024 * 1024 / 16;
    f4* src; float* sink;
    CHECK(hipMalloc(&src, nvec * 16)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 0, nvec * 16));
    printf("%zu MiB\n", mib);
#define R(T, U, TL, NT, ST, PF) run<T, U, TL, NT, ST, PF>("threads " #T " U " #U " tiles " #TL " nt " #NT " strided " #ST " prefetch " #PF, src, nvec, sink)
    R(256, 4, 8, true, false, false);
    R(256, 4, 8, false, false, false);
    R(256, 4, 8, true, false, true);
    R(256, 4, 8, false, false, true);
    R(256, 4, 8, true, true, false);
    R(256, 4, 8, false, true, true);
    R(256, 4, 1, true, false, false);
    R(256, 4, 2, true, false, false);
    R(256, 4, 4, true, false, false);
    R(256, 4, 16, true, false, false);
    R(256, 4, 32, true, false, true);
    R(256, 8, 4, true, false, false);
    R(256, 8, 1, true, false, false);
    R(256, 2, 8, true, false, false);
    R(256, 1, 16, true, false, false);
    R(512, 4, 4, true, false, false);
    R(1024, 4, 2, true, false, false);
    R(1024, 4, 8, true, false, true);
    R(64, 4, 16, true, false, false);
    R(64, 8, 16, true, false, true);
    return 0;
}
