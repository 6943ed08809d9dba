// Micro-benchmark: store patterns and arithmetic variants for the N x M pairwise IoU matrix (write-bound: 4*N*M bytes).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pairwise_bw pairwise_bw.hip && ./pairwise_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// ARITH: 0 = store a constant, 1 = IoU with IEEE division, 2 = IoU with inter * rcp(union) (not bit-exact; ALU probe)
template <int ARITH>
__device__ __forceinline__ float iou(const float4 r, float ra, const float4 c, float ca)
{
    if (ARITH == 0) return ra;
    const float inter = fmaxf(fminf(r.z, c.z) - fmaxf(r.x, c.x), 0.0f) * fmaxf(fminf(r.w, c.w) - fmaxf(r.y, c.y), 0.0f);
    if (ARITH == 2) return inter * __builtin_amdgcn_rcpf(ra + ca - inter);
    return __fdiv_rn(inter, ra + ca - inter);
}

// tile: ROWS rows x (64*4) columns per wave-row; WAVES waves; INTERLEAVE: wave w takes rows w, w+WAVES, ... else a block of rows
template <int ROWS, int WAVES, int ARITH, int NT, int INTERLEAVE>
__global__ void __launch_bounds__(WAVES * 64) k_tile(const float4* __restrict__ a, const float4* __restrict__ b, float* __restrict__ out,
                                                     int N, int M, int col_tiles)
{
    __shared__ float4 rows[ROWS];
    __shared__ float rarea[ROWS];
    const int rt = blockIdx.x / col_tiles, ct = blockIdx.x - rt * col_tiles;
    const long row0 = (long)rt * ROWS;
    for (int t = threadIdx.x; t < ROWS; t += WAVES * 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + t < N) v = a[row0 + t];
        rows[t] = v;
        rarea[t] = (v.z - v.x) * (v.w - v.y);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = (N - row0) < ROWS ? (int)(N - row0) : ROWS;
    const long col = (long)ct * 256 + lane * 4;
    if (col >= M) return;
    float4 c[4];
    float ca[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = b[col + j]; ca[j] = (c[j].z - c[j].x) * (c[j].w - c[j].y); }
    float* __restrict__ o = out + (size_t)row0 * M + col;
    const int per = ROWS / WAVES;
    const int k0 = INTERLEAVE ? wave : wave * per, step = INTERLEAVE ? WAVES : 1, kend = INTERLEAVE ? nrows : min(nrows, (wave + 1) * per);
#pragma unroll 2
    for (int k = k0; k < kend; k += step) {
        const float4 r = rows[k];
        const float ra = rarea[k];
        f4 v;
        v.x = iou<ARITH>(r, ra, c[0], ca[0]); v.y = iou<ARITH>(r, ra, c[1], ca[1]);
        v.z = iou<ARITH>(r, ra, c[2], ca[2]); v.w = iou<ARITH>(r, ra, c[3], ca[3]);
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(o + (size_t)k * M));
        else *reinterpret_cast<f4*>(o + (size_t)k * M) = v;
    }
}

// flat: a thread owns U vectors of 4 consecutive outputs, workgroup-contiguous like a plain stream; boxes come from the caches
template <int U, int ARITH>
__global__ void __launch_bounds__(256) k_flat(const float4* __restrict__ a, const float4* __restrict__ b, float* __restrict__ out,
                                              int N, int M)
{
    const size_t total = (size_t)N * M / 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t v = (size_t)blockIdx.x * (256 * U) + u * 256 + threadIdx.x;
        if (v >= total) continue;
        const size_t e = v * 4;
        const unsigned row = (unsigned)(e / (unsigned)M), col = (unsigned)(e - (size_t)row * M);
        const float4 r = a[row];
        const float ra = (r.z - r.x) * (r.w - r.y);
        f4 o;
        float4 c;
        c = b[col]; o.x = iou<ARITH>(r, ra, c, (c.z - c.x) * (c.w - c.y));
        c = b[col + 1]; o.y = iou<ARITH>(r, ra, c, (c.z - c.x) * (c.w - c.y));
        c = b[col + 2]; o.z = iou<ARITH>(r, ra, c, (c.z - c.x) * (c.w - c.y));
        c = b[col + 3]; o.w = iou<ARITH>(r, ra, c, (c.z - c.x) * (c.w - c.y));
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out + e));
    }
}

// row strips: a workgroup owns ROWS rows x 1024 columns: a lane keeps 4 column boxes per 256-column piece (PIECES of them)
template <int ROWS, int PIECES, int ARITH>
__global__ void __launch_bounds__(256) k_strip(const float4* __restrict__ a, const float4* __restrict__ b, float* __restrict__ out,
                                               int N, int M, int col_tiles)
{
    __shared__ float4 rows[ROWS];
    __shared__ float rarea[ROWS];
    const int rt = blockIdx.x / col_tiles, ct = blockIdx.x - rt * col_tiles;
    const long row0 = (long)rt * ROWS;
    for (int t = threadIdx.x; t < ROWS; t += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + t < N) v = a[row0 + t];
        rows[t] = v;
        rarea[t] = (v.z - v.x) * (v.w - v.y);
    }
    __syncthreads();
    const int nrows = (N - row0) < ROWS ? (int)(N - row0) : ROWS;
    // the workgroup's 256 lanes x 4 columns = 1024 contiguous columns per row and piece
    for (int p = 0; p < PIECES; ++p) {
        const long col = ((long)ct * PIECES + p) * 1024 + threadIdx.x * 4;
        if (col >= M) continue;
        float4 c[4];
        float ca[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = b[col + j]; ca[j] = (c[j].z - c[j].x) * (c[j].w - c[j].y); }
        float* __restrict__ o = out + (size_t)row0 * M + col;
#pragma unroll 2
        for (int k = 0; k < nrows; ++k) {
            const float4 r = rows[k];
            const float ra = rarea[k];
            f4 v;
            v.x = iou<ARITH>(r, ra, c[0], ca[0]); v.y = iou<ARITH>(r, ra, c[1], ca[1]);
            v.z = iou<ARITH>(r, ra, c[2], ca[2]); v.w = iou<ARITH>(r, ra, c[3], ca[3]);
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(o + (size_t)k * M));
        }
    }
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 8192, M = argc > 2 ? atoi(argv[2]) : 8192;
    const size_t bytes = (size_t)N * M * 4;
    float4 *a, *b;
    float* out;
    CHECK(hipMalloc(&a, (size_t)N * 16)); CHECK(hipMalloc(&b, (size_t)M * 16)); CHECK(hipMalloc(&out, bytes));
    float4* h = (float4*)malloc((size_t)(N > M ? N : M) * 16);
    srand(1);
    for (int i = 0; i < (N > M ? N : M); ++i) {
        const float x = rand() % 600, y = rand() % 600;
        h[i] = make_float4(x, y, x + 1 + rand() % 200, y + 1 + rand() % 200);
    }
    CHECK(hipMemcpy(a, h, (size_t)N * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, h, (size_t)M * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %8.1f us  %7.1f GB/s\n", name, ms / 20 * 1e3, (double)bytes / (ms / 20 * 1e-3) / 1e9);
    };
    printf("N = %d, M = %d, %.1f MB out\n", N, M, bytes / 1e6);
    const int ct = (M + 255) / 256;
#define TILE(ROWS, WAVES, ARITH, NT, IL) timeit("tile rows " #ROWS " waves " #WAVES " arith " #ARITH " nt " #NT " interleave " #IL, [&] { \
        hipLaunchKernelGGL((k_tile<ROWS, WAVES, ARITH, NT, IL>), dim3((unsigned)(((N + ROWS - 1) / ROWS) * ct)), dim3(WAVES * 64), 0, 0, a, b, out, N, M, ct); })
    TILE(32, 4, 1, 1, 1); TILE(32, 4, 0, 1, 1); TILE(32, 4, 2, 1, 1); TILE(32, 4, 1, 0, 1); TILE(32, 4, 1, 1, 0); TILE(32, 4, 0, 1, 0);
    TILE(16, 4, 1, 1, 1); TILE(64, 4, 1, 1, 1); TILE(128, 4, 1, 1, 1); TILE(8, 1, 1, 1, 1); TILE(16, 1, 1, 1, 1); TILE(32, 1, 1, 1, 1); TILE(32, 1, 0, 1, 1);
    TILE(16, 2, 1, 1, 1); TILE(32, 2, 1, 1, 1); TILE(8, 4, 1, 1, 1); TILE(4, 4, 1, 1, 1); TILE(4, 4, 0, 1, 1);
    if ((size_t)N * M % 4 == 0 && M % 4 == 0) {
#define FLAT(U, ARITH) timeit("flat unroll " #U " arith " #ARITH, [&] { \
        hipLaunchKernelGGL((k_flat<U, ARITH>), dim3((unsigned)(((size_t)N * M / 4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, a, b, out, N, M); })
        FLAT(1, 1); FLAT(1, 0); FLAT(2, 1); FLAT(1, 2);
    }
#define STRIP(ROWS, PIECES, ARITH) timeit("strip rows " #ROWS " x 1024 columns x pieces " #PIECES " arith " #ARITH, [&] { \
        const int sct = (M + 1024 * PIECES - 1) / (1024 * PIECES); \
        hipLaunchKernelGGL((k_strip<ROWS, PIECES, ARITH>), dim3((unsigned)(((N + ROWS - 1) / ROWS) * sct)), dim3(256), 0, 0, a, b, out, N, M, sct); })
    STRIP(8, 1, 1); STRIP(8, 1, 0); STRIP(16, 1, 1); STRIP(4, 1, 1); STRIP(4, 2, 1); STRIP(8, 8, 1); STRIP(2, 8, 1); STRIP(1, 8, 1); STRIP(1, 8, 0);
    return 0;
}
