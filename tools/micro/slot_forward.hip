// What would it save to drop the target-assignment launch from the YOLO loss?  Measured, not estimated.
//
// Today (path A, the library): k_build_targets (assignment of all levels, order-preserving compaction; its spare
// workgroups set the per-cell map to -1) -> k_yl_pos_fwd (one wave per assigned row).
// Alternative (path B, this file): a fill launch (per-cell map = -1, counters = 0) -> k_slot_fwd, one wave per
// (level, offset, anchor, target) SLOT.  A wave tests its own slot (anchor ratio, neighbour offset: loss.py:191-203),
// leaves when the reference would not have produced a row for it, and otherwise does what the positives' forward does,
// with the slot number as row id (slot numbers grow in the reference's row order, so "last row of a cell" survives).
// The file checks that B yields the rows of A (counts per level, IoU and class sum of every row, in order) and times
// both pairs of launches back to back on one stream; run it under rocprofv3 --kernel-trace for the kernel durations.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../object_detection_amd/csrc -I../../include \
//         -L../../object_detection_amd -ldethead -Wl,-rpath,'$ORIGIN/../../object_detection_amd' -o slot_forward slot_forward.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dh_iou.h"
#include "dethead.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CHECK_DH(x) do { int e_ = (x); if (e_ != 0) { printf("%s: %d\n", #x, e_); exit(1); } } while (0)

#define NL 3

struct slot_params {
    const float* p[NL];
    int ny[NL], nx[NL];
    long cell_first[NL];
    int bs, na, nc, no, nt;
    float anchor_t, cp, cn, cls_pw;
    long cap;      // slots per level = 5*na*nt
};

__device__ __forceinline__ float bce_plain(float x, float t, float pw)
{
    const float lw = (pw - 1.0f) * t + 1.0f;
    return (1.0f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
}

__device__ __forceinline__ float remainder_one(float x)
{
    float m = x - truncf(x);
    if (m != 0.0f && m < 0.0f) m += 1.0f;
    return m;
}

// IoU and class sum of one row, by one wave (lanes over classes)
__device__ __forceinline__ void row_forward(const slot_params& P, const float* __restrict__ ps, const float tb[4], const float an[2],
                                            int cls, int lane, float& iou, float& csum)
{
    const float x0 = ps[0], x1 = ps[1], x2 = ps[2], x3 = ps[3];
    const float px = dh_sigmoid(x0) * 2.0f - 0.5f, py = dh_sigmoid(x1) * 2.0f - 0.5f;
    const float tw = dh_sigmoid(x2) * 2.0f, th = dh_sigmoid(x3) * 2.0f;
    const float pw = (tw * tw) * an[0], ph = (th * th) * an[1];
    iou = dh_iou_value<DH_CIOU>(dh_from_xywh(px, py, pw, ph), dh_from_xywh(tb[0], tb[1], tb[2], tb[3]), 1e-7f, (float)(1.0 + 1e-7));
    csum = 0.0f;
    for (int c = lane; c < P.nc; c += 64) csum += bce_plain(ps[5 + c], c == cls ? P.cp : P.cn, P.cls_pw);
    csum = dh_wave_sum(csum);
}

__global__ void __launch_bounds__(256) k_fill_map(uint4* __restrict__ map, size_t vecs, int* __restrict__ counts)
{
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    if (i < vecs) map[i] = ones;
    if (i + 256 < vecs) map[i + 256] = ones;
    if (blockIdx.x == 0 && threadIdx.x < NL) counts[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) k_slot_fwd(const slot_params P, const float* __restrict__ targets,
                                                  const float* __restrict__ anchors, int* __restrict__ owner,
                                                  int* __restrict__ nxt, float* __restrict__ iou_ws, float* __restrict__ cls_ws,
                                                  int* __restrict__ counts)
{
    const int l = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= P.cap) return;
    const int nx = P.nx[l], ny = P.ny[l];
    const float fnx = (float)nx, fny = (float)ny;
    const int per_off = P.na * P.nt;
    const int off = (int)(s / per_off), q = (int)(s - (long)off * per_off);
    const int a = q / P.nt, t = q - a * P.nt;
    const float* __restrict__ tr = targets + (size_t)t * 6;
    const float an[2] = {anchors[((size_t)l * P.na + a) * 2], anchors[((size_t)l * P.na + a) * 2 + 1]};
    const float gw = tr[4] * fnx, gh = tr[5] * fny;
    const float rw = __fdiv_rn(gw, an[0]), rh = __fdiv_rn(gh, an[1]);
    const float worst = fmaxf(fmaxf(rw, __fdiv_rn(1.0f, rw)), fmaxf(rh, __fdiv_rn(1.0f, rh)));
    const float gx = tr[2] * fnx, gy = tr[3] * fny;
    bool take = worst < P.anchor_t;
    if (off == 1) take = take && (remainder_one(gx) < 0.5f) && (gx > 1.0f);
    if (off == 2) take = take && (remainder_one(gy) < 0.5f) && (gy > 1.0f);
    if (off == 3) { const float ix = fnx - gx; take = take && (remainder_one(ix) < 0.5f) && (ix > 1.0f); }
    if (off == 4) { const float iy = fny - gy; take = take && (remainder_one(iy) < 0.5f) && (iy > 1.0f); }
    float* __restrict__ iou_out = iou_ws + (size_t)l * P.cap;
    float* __restrict__ cls_out = cls_ws + (size_t)l * P.cap;
    if (!take) {      // not a row of the reference: contributes 1 - iou = 0 and no class term
        if (lane == 0) { iou_out[s] = 1.0f; cls_out[s] = 0.0f; nxt[(size_t)l * P.cap + s] = -2; }
        return;
    }
    const float ox = off == 1 ? 0.5f : (off == 3 ? -0.5f : 0.0f);
    const float oy = off == 2 ? 0.5f : (off == 4 ? -0.5f : 0.0f);
    long long gi = (long long)(gx - ox), gj = (long long)(gy - oy);
    gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);
    gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);
    const long long b = (long long)tr[0];
    if (b < 0 || b >= P.bs) {
        if (lane == 0) { iou_out[s] = 1.0f; cls_out[s] = 0.0f; nxt[(size_t)l * P.cap + s] = -2; }
        return;
    }
    const float tb[4] = {gx - (float)gi, gy - (float)gj, gw, gh};
    const long cell = ((b * P.na + a) * ny + gj) * nx + gi;
    float iou, csum;
    row_forward(P, P.p[l] + (size_t)cell * P.no, tb, an, (int)tr[1], lane, iou, csum);
    if (lane == 0) {
        iou_out[s] = iou;
        cls_out[s] = csum;
        nxt[(size_t)l * P.cap + s] = atomicExch(owner + P.cell_first[l] + cell, (int)s);
    }
}

// Path C: a wave takes 16 slots; every lane tests slot (lane & 15), and the rows that exist are worked on four at a time,
// one per group of 16 lanes.  A lane of a group stands for four of the 64 lanes of the row-per-wave kernel (lane l for l, l+16,
// l+32, l+48): it forms their class sums and the first two butterfly steps of the wave reduction by itself, and the last four
// steps run across the group -- the same additions in the same tree, so the class sum has the same bits.
// INTERLEAVE: the 16 slots of a wave lie n_waves apart (every wave sees every offset and anchor) instead of side by side.
template <bool INTERLEAVE, int G>
__global__ void __launch_bounds__(256) k_slot_fwd16(const slot_params P, const float* __restrict__ targets,
                                                    const float* __restrict__ anchors, int* __restrict__ owner,
                                                    int* __restrict__ nxt, float* __restrict__ iou_ws, float* __restrict__ cls_ws)
{
    const int l = blockIdx.y;
    constexpr int U = 64 / G;          // lanes of the row-per-wave kernel one lane stands for = rows a wave works on at a time
    const int lane = threadIdx.x & 63, slot_lane = lane & 15, sub = lane & (G - 1), grp = lane / G;
    const long n_waves = (P.cap + 15) / 16;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const long s = INTERLEAVE ? w + (long)slot_lane * n_waves : w * 16 + slot_lane;
    const int nx = P.nx[l], ny = P.ny[l];
    const float fnx = (float)nx, fny = (float)ny;
    bool take = s < P.cap;
    long cell = 0;
    float tb0 = 0.f, tb1 = 0.f, tb2 = 0.f, tb3 = 0.f, an0 = 1.f, an1 = 1.f;
    int cls = 0;
    if (take) {
        const int per_off = P.na * P.nt;
        const int off = (int)(s / per_off), q = (int)(s - (long)off * per_off);
        const int a = q / P.nt, t = q - a * P.nt;
        const float* __restrict__ tr = targets + (size_t)t * 6;
        an0 = anchors[((size_t)l * P.na + a) * 2]; an1 = anchors[((size_t)l * P.na + a) * 2 + 1];
        const float gw = tr[4] * fnx, gh = tr[5] * fny;
        const float rw = __fdiv_rn(gw, an0), rh = __fdiv_rn(gh, an1);
        const float worst = fmaxf(fmaxf(rw, __fdiv_rn(1.0f, rw)), fmaxf(rh, __fdiv_rn(1.0f, rh)));
        const float gx = tr[2] * fnx, gy = tr[3] * fny;
        take = worst < P.anchor_t;
        if (off == 1) take = take && (remainder_one(gx) < 0.5f) && (gx > 1.0f);
        if (off == 2) take = take && (remainder_one(gy) < 0.5f) && (gy > 1.0f);
        if (off == 3) { const float ix = fnx - gx; take = take && (remainder_one(ix) < 0.5f) && (ix > 1.0f); }
        if (off == 4) { const float iy = fny - gy; take = take && (remainder_one(iy) < 0.5f) && (iy > 1.0f); }
        const float ox = off == 1 ? 0.5f : (off == 3 ? -0.5f : 0.0f);
        const float oy = off == 2 ? 0.5f : (off == 4 ? -0.5f : 0.0f);
        long long gi = (long long)(gx - ox), gj = (long long)(gy - oy);
        gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);
        gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);
        const long long b = (long long)tr[0];
        take = take && b >= 0 && b < P.bs;
        tb0 = gx - (float)gi; tb1 = gy - (float)gj; tb2 = gw; tb3 = gh;
        cell = ((b * P.na + a) * ny + gj) * nx + gi;
        cls = (int)tr[1];
    }
    float* __restrict__ iou_out = iou_ws + (size_t)l * P.cap;
    float* __restrict__ cls_out = cls_ws + (size_t)l * P.cap;
    int* __restrict__ nxt_l = nxt + (size_t)l * P.cap;
    if (!take && lane < 16 && s < P.cap) { iou_out[s] = 1.0f; cls_out[s] = 0.0f; nxt_l[s] = -2; }
    const unsigned mask = (unsigned)(__ballot(take) & 0xFFFFull);
    const int count = __popc(mask);
    for (int pass = 0; pass * U < count; ++pass) {
        const int k = pass * U + grp;
        const bool active = k < count;
        unsigned m = mask;
        for (int j = 0; j < k && j < 15; ++j) m &= m - 1;
        const int src = active ? __ffs(m) - 1 : 0;
        const long r_cell = __shfl(cell, src, 64);
        const long r_s = __shfl(s, src, 64);
        const float r_tb[4] = {__shfl(tb0, src, 64), __shfl(tb1, src, 64), __shfl(tb2, src, 64), __shfl(tb3, src, 64)};
        const float r_an[2] = {__shfl(an0, src, 64), __shfl(an1, src, 64)};
        const int r_cls = __shfl(cls, src, 64);
        float iou = 0.0f, s16 = 0.0f;
        if (active) {
            const float* __restrict__ ps = P.p[l] + (size_t)r_cell * P.no;
            const float x0 = ps[0], x1 = ps[1], x2 = ps[2], x3 = ps[3];
            float V[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                V[u] = 0.0f;
                for (int c = sub + G * u; c < P.nc; c += 64) V[u] += bce_plain(ps[5 + c], c == r_cls ? P.cp : P.cn, P.cls_pw);
            }
            const float px = dh_sigmoid(x0) * 2.0f - 0.5f, py = dh_sigmoid(x1) * 2.0f - 0.5f;
            const float tw = dh_sigmoid(x2) * 2.0f, th = dh_sigmoid(x3) * 2.0f;
            const float pw = (tw * tw) * r_an[0], ph = (th * th) * r_an[1];
            iou = dh_iou_value<DH_CIOU>(dh_from_xywh(px, py, pw, ph), dh_from_xywh(r_tb[0], r_tb[1], r_tb[2], r_tb[3]), 1e-7f, (float)(1.0 + 1e-7));
#pragma unroll
            for (int du = U / 2; du > 0; du >>= 1)       // the butterfly steps 32 ... G of the 64-lane reduction, inside the lane
#pragma unroll
                for (int u = 0; u < du; ++u) V[u] = V[u] + V[u + du];
            s16 = V[0];
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s16 += __shfl_xor(s16, o, 64);
        if (active && sub == 0) {
            iou_out[r_s] = iou;
            cls_out[r_s] = s16;
            nxt_l[r_s] = atomicExch(owner + P.cell_first[l] + r_cell, (int)r_s);
        }
    }
}

// the same row arithmetic on the rows the library's assignment produced (the check of path B)
__global__ void __launch_bounds__(256) k_rows_check(const slot_params P, const long long* __restrict__ idx,
                                                    const float* __restrict__ tbox, const float* __restrict__ anch,
                                                    const long long* __restrict__ tcls, const int* __restrict__ counts,
                                                    float* __restrict__ iou_ws, float* __restrict__ cls_ws)
{
    const int l = blockIdx.y, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= counts[l]) return;
    const long long* base = idx + (size_t)l * 4 * P.cap;
    const long long b = base[r], a = base[P.cap + r], gj = base[2 * P.cap + r], gi = base[3 * P.cap + r];
    const long cell = ((b * P.na + a) * P.ny[l] + gj) * P.nx[l] + gi;
    const float* tbp = tbox + ((size_t)l * P.cap + r) * 4;
    const float tb[4] = {tbp[0], tbp[1], tbp[2], tbp[3]};
    const float an[2] = {anch[((size_t)l * P.cap + r) * 2], anch[((size_t)l * P.cap + r) * 2 + 1]};
    float iou, csum;
    row_forward(P, P.p[l] + (size_t)cell * P.no, tb, an, (int)tcls[(size_t)l * P.cap + r], lane, iou, csum);
    if (lane == 0) { iou_ws[(size_t)l * P.cap + r] = iou; cls_ws[(size_t)l * P.cap + r] = csum; }
}

__global__ void k_random(float* p, size_t n, unsigned seed, float lo, float hi)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
    }
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
    auto path_a = [&]() {
        CHECK_DH(dh_yolo_loss_positives(d_t, nt, d_a, 4.0, pp, ny, nx, NL, bs, na, nc, (int64_t*)idx, tbox, anch, (int64_t*)tcls,
                                        counts_a, cap, &hyp, aws, aws_bytes, ws, ws_bytes, st));
    };
    // path B buffers
    int *owner, *nxt, *counts_b; float *iou_b, *cls_b, *iou_c, *cls_c;
    const size_t map_bytes = ((size_t)cells_total * 4 + 15) / 16 * 16;
    CHECK(hipMalloc(&owner, map_bytes)); CHECK(hipMalloc(&nxt, (size_t)NL * cap * 4)); CHECK(hipMalloc(&counts_b, 64));
    CHECK(hipMalloc(&iou_b, (size_t)NL * cap * 4)); CHECK(hipMalloc(&cls_b, (size_t)NL * cap * 4));
    CHECK(hipMalloc(&iou_c, (size_t)NL * cap * 4)); CHECK(hipMalloc(&cls_c, (size_t)NL * cap * 4));
    const size_t vecs = map_bytes / 16;
    auto path_b = [&]() {
        hipLaunchKernelGGL(k_fill_map, dim3((unsigned)((vecs + 511) / 512)), dim3(256), 0, st, (uint4*)owner, vecs, counts_b);
        hipLaunchKernelGGL(k_slot_fwd, dim3((unsigned)((cap + 3) / 4), NL), dim3(256), 0, st, P, d_t, d_a, owner, nxt, iou_b, cls_b, counts_b);
    };

    const long n_waves16 = (cap + 15) / 16;
    int lanes_per_row = 16;
    auto path_c = [&](bool interleave) {
        hipLaunchKernelGGL(k_fill_map, dim3((unsigned)((vecs + 511) / 512)), dim3(256), 0, st, (uint4*)owner, vecs, counts_b);
        const dim3 grid((unsigned)((n_waves16 + 3) / 4), NL);
#define SLOT16(IL, G) hipLaunchKernelGGL((k_slot_fwd16<IL, G>), grid, dim3(256), 0, st, P, d_t, d_a, owner, nxt, iou_b, cls_b)
        if (lanes_per_row == 16) { if (interleave) SLOT16(true, 16); else SLOT16(false, 16); }
        else if (lanes_per_row == 8) { if (interleave) SLOT16(true, 8); else SLOT16(false, 8); }
        else { if (interleave) SLOT16(true, 4); else SLOT16(false, 4); }
#undef SLOT16
    };
    // ---- check: B's valid slots, in slot order, are A's rows
    long bad_total = 0;
    for (int variant = 0; variant < 7; ++variant) {
    path_a();
    lanes_per_row = variant < 3 ? 16 : (variant < 5 ? 8 : 4);
    if (variant == 0) path_b(); else path_c(variant % 2 == 0);
    hipLaunchKernelGGL(k_rows_check, dim3((unsigned)((cap + 3) / 4), NL), dim3(256), 0, st, P, idx, tbox, anch, tcls, counts_a, iou_c, cls_c);
    CHECK(hipStreamSynchronize(st));
    int ca[NL], cb[NL];
    CHECK(hipMemcpy(ca, counts_a, sizeof ca, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(cb, counts_b, sizeof cb, hipMemcpyDeviceToHost));
    std::vector<float> hb((size_t)NL * cap), hcb((size_t)NL * cap), hc((size_t)NL * cap), hcc((size_t)NL * cap);
    std::vector<int> hn((size_t)NL * cap);
    CHECK(hipMemcpy(hb.data(), iou_b, hb.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hcb.data(), cls_b, hb.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hc.data(), iou_c, hb.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hcc.data(), cls_c, hb.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hn.data(), nxt, hn.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0, total = 0;
    for (int l = 0; l < NL; ++l) {
        long r = 0;
        for (long s = 0; s < cap; ++s) {
            if (hn[(size_t)l * cap + s] == -2) continue;
            if (r < ca[l] && (memcmp(&hb[(size_t)l * cap + s], &hc[(size_t)l * cap + r], 4) || memcmp(&hcb[(size_t)l * cap + s], &hcc[(size_t)l * cap + r], 4))) ++bad;
            ++r;
        }
        cb[l] = (int)r;      // (a real implementation counts the rows in the reduction that follows: 9 000 atomic adds on three
                             // words took this kernel from 11 to 108 us)
        if (r != ca[l]) ++bad;
        total += ca[l];
        if (variant == 0) printf("level %d: rows %d (library) %d (slots) of %ld slots\n", l, ca[l], cb[l], cap);
    }
    if (variant == 0) printf("B (wave per slot): rows %ld, mismatches %ld\n", total, bad);
    else printf("C (16 slots per wave, %d lanes per row, %s): rows %ld, mismatches %ld\n", lanes_per_row, variant % 2 == 0 ? "interleaved" : "side by side", total, bad);
    bad_total += bad;
    }

    // ---- timing, back to back on one stream
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 20; ++i) path_a();
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) path_a();
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("A  assignment+fill -> positives' forward (library)   %7.2f us per pair of launches\n", ms * 1000.0f / reps);
        for (int i = 0; i < 20; ++i) path_b();
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) path_b();
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("B  fill -> one wave per slot                          %7.2f us per pair of launches\n", ms * 1000.0f / reps);
        for (int il = 0; il < 6; ++il) {
            lanes_per_row = il < 2 ? 16 : (il < 4 ? 8 : 4);
            for (int i = 0; i < 20; ++i) path_c(il % 2 != 0);
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) path_c(il % 2 != 0);
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("C  fill -> 16 slots per wave, %2d lanes per row (%s) %7.2f us per pair of launches\n", lanes_per_row, il % 2 ? "interleaved" : "side by side", ms * 1000.0f / reps);
        }
    }
    return bad_total != 0;
}
