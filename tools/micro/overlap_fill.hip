// How well do the loss's short launches hide beside a bandwidth-bound fill on a second stream, and does the shape of the
// fill matter?  Main stream: the library's assignment + positives' forward, twice (four latency-bound launches, ~40 us).
// Second stream: 548 MB of zeros, either one 16-byte store per thread (as many workgroups as it takes) or a fixed number
// of workgroups that loop (fewer waves resident: compute-unit slots stay free for the other stream's workgroups).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../object_detection_amd/csrc -I../../include \
//         -L../../object_detection_amd -ldethead -Wl,-rpath,'$ORIGIN/../../object_detection_amd' -o overlap_fill overlap_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dh_common.h"
#include "dethead.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CHECK_DH(x) do { int e_ = (x); if (e_ != 0) { printf("%s: %d\n", #x, e_); exit(1); } } while (0)
#define NL 3

struct slot_params { const float* p[NL]; int ny[NL], nx[NL]; long cell_first[NL]; int bs, na, nc, no, nt; float anchor_t, cp, cn, cls_pw; long cap; };

__global__ void k_random(float* p, size_t n, unsigned seed, float lo, float hi)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
    }
}

__global__ void __launch_bounds__(256) k_fill_once(dh_f4* __restrict__ p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const dh_f4 z = {0.f, 0.f, 0.f, 0.f};
    if (i < n) __builtin_nontemporal_store(z, p + i);
}

// THREADS threads per workgroup, gridDim.x workgroups that stride over the buffer
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_fill_loop(dh_f4* __restrict__ p, size_t n)
{
    const dh_f4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * THREADS) __builtin_nontemporal_store(z, p + i);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    const int bs = 64, na = 3, nc = 80, no = 85, nt = 640;
    const int ny[NL] = {80, 40, 20}, nx[NL] = {80, 40, 20};
    const float stride[NL] = {8.f, 16.f, 32.f};
    const float anchor_px[NL][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};
    const long cap = 5L * na * nt;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));

    slot_params P;
    memset(&P, 0, sizeof P);
    float* maps[NL];
    long cells_total = 0;
    for (int l = 0; l < NL; ++l) {
        const size_t n = (size_t)bs * na * ny[l] * nx[l] * no;
        CHECK(hipMalloc(&maps[l], n * 4));
        hipLaunchKernelGGL(k_random, dim3(4096), dim3(256), 0, st, maps[l], n, 17u + l, -4.0f, 4.0f);
        P.p[l] = maps[l]; P.ny[l] = ny[l]; P.nx[l] = nx[l];
        P.cell_first[l] = cells_total;
        cells_total += (long)bs * na * ny[l] * nx[l];
    }
    P.bs = bs; P.na = na; P.nc = nc; P.no = no; P.nt = nt; P.anchor_t = 4.0f; P.cp = 1.0f; P.cn = 0.0f; P.cls_pw = 1.0f; P.cap = cap;

    // targets: image (sorted), class, centre, size -- sizes around the anchors so that the ratio test passes for some of them
    std::vector<float> h_t((size_t)nt * 6), h_a(NL * na * 2);
    unsigned rs = 12345u;
    auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return (float)(rs >> 8) * (1.0f / 16777216.0f); };
    for (int t = 0; t < nt; ++t) {
        const int l = t % NL, a = (t / NL) % na;
        h_t[6 * t + 0] = (float)(t * bs / nt);
        h_t[6 * t + 1] = (float)(int)(rnd() * nc);
        h_t[6 * t + 2] = 0.02f + 0.96f * rnd();
        h_t[6 * t + 3] = 0.02f + 0.96f * rnd();
        h_t[6 * t + 4] = anchor_px[l][2 * a] * (0.6f + 0.9f * rnd()) / 640.0f;
        h_t[6 * t + 5] = anchor_px[l][2 * a + 1] * (0.6f + 0.9f * rnd()) / 640.0f;
    }
    for (int l = 0; l < NL; ++l) for (int k = 0; k < na * 2; ++k) h_a[l * na * 2 + k] = anchor_px[l][k] / stride[l];
    float *d_t, *d_a;
    CHECK(hipMalloc(&d_t, h_t.size() * 4)); CHECK(hipMalloc(&d_a, h_a.size() * 4));
    CHECK(hipMemcpy(d_t, h_t.data(), h_t.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_a, h_a.data(), h_a.size() * 4, hipMemcpyHostToDevice));

    // path A buffers
    long long *idx, *tcls; float *tbox, *anch; int* counts_a;
    CHECK(hipMalloc(&idx, (size_t)NL * 4 * cap * 8)); CHECK(hipMalloc(&tcls, (size_t)NL * cap * 8));
    CHECK(hipMalloc(&tbox, (size_t)NL * cap * 16)); CHECK(hipMalloc(&anch, (size_t)NL * cap * 8)); CHECK(hipMalloc(&counts_a, 64));
    const size_t aws_bytes = dh_build_targets_workspace(NL, na, nt), ws_bytes = dh_yolo_loss_workspace(ny, nx, NL, bs, na, nc, cap);
    void *aws, *ws;
    CHECK(hipMalloc(&aws, aws_bytes)); CHECK(hipMalloc(&ws, ws_bytes));
    dh_yolo_hyp hyp;
    memset(&hyp, 0, sizeof hyp);
    hyp.box = 0.05f; hyp.obj = 1.0f; hyp.cls = 0.5f; hyp.cls_pw = 1.0f; hyp.obj_pw = 1.0f; hyp.cp = 1.0f; hyp.cn = 0.0f; hyp.gr = 1.0f;
    hyp.balance[0] = 4.0f; hyp.balance[1] = 1.0f; hyp.balance[2] = 0.4f;
    const float* pp[NL] = {maps[0], maps[1], maps[2]};

    auto forward = [&]() {
        for (int k = 0; k < 2; ++k)
            CHECK_DH(dh_yolo_loss_positives(d_t, nt, d_a, 4.0, pp, ny, nx, NL, bs, na, nc, (int64_t*)idx, tbox, anch, (int64_t*)tcls,
                                            counts_a, cap, &hyp, aws, aws_bytes, ws, ws_bytes, st));
    };
    size_t fill_bytes = 0;
    for (int l = 0; l < NL; ++l) fill_bytes += (size_t)bs * na * ny[l] * nx[l] * no * 4;
    dh_f4* grad;
    CHECK(hipMalloc(&grad, fill_bytes));
    const size_t nvec = fill_bytes / 16;
    hipStream_t side;
    CHECK(hipStreamCreate(&side));
    hipEvent_t fork, join, e0, e1;
    CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct shape { const char* name; int threads; unsigned blocks; };
    const shape shapes[] = {{"one store per thread", 0, 0}, {"loop, 4096 x 256", 256, 4096}, {"loop, 2048 x 256", 256, 2048},
                            {"loop, 1024 x 256", 256, 1024}, {"loop, 512 x 256", 256, 512}, {"loop, 1024 x 64", 64, 1024},
                            {"loop, 512 x 64", 64, 512}, {"loop, 256 x 1024", 1024, 256}};
    auto fill = [&](const shape& s, hipStream_t on) {
        if (s.threads == 0) hipLaunchKernelGGL(k_fill_once, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, on, grad, nvec);
        else if (s.threads == 64) hipLaunchKernelGGL(k_fill_loop<64>, dim3(s.blocks), dim3(64), 0, on, grad, nvec);
        else if (s.threads == 1024) hipLaunchKernelGGL(k_fill_loop<1024>, dim3(s.blocks), dim3(1024), 0, on, grad, nvec);
        else hipLaunchKernelGGL(k_fill_loop<256>, dim3(s.blocks), dim3(256), 0, on, grad, nvec);
    };
    auto timed = [&](auto&& body) {
        for (int i = 0; i < 50; ++i) body();
        CHECK(hipStreamSynchronize(st)); CHECK(hipStreamSynchronize(side));
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) body();
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0f / reps;
    };
    const float t_fwd = timed([&]() { forward(); });
    printf("four short launches alone (assignment, positives' forward, twice)   %7.1f us\n", t_fwd);
    for (const shape& s : shapes) {
        const float t_fill = timed([&]() { fill(s, st); });
        const float t_seq = timed([&]() { fill(s, st); forward(); });
        const float t_par = timed([&]() {
            CHECK(hipEventRecord(fork, st)); CHECK(hipStreamWaitEvent(side, fork, 0));
            fill(s, side);
            CHECK(hipEventRecord(join, side));
            forward();
            CHECK(hipStreamWaitEvent(st, join, 0));
        });
        printf("fill %-22s alone %6.1f us (%4.2f TB/s)   then the launches %6.1f us   beside them %6.1f us   hidden %5.1f us of %4.1f\n",
               s.name, t_fill, fill_bytes / t_fill / 1e6, t_seq, t_par, t_seq - t_par, t_fwd);
    }
    return 0;
}
