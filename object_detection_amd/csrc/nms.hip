// Batched class-aware NMS + top-k for YOLOv5-style predictions, and a plain nms(boxes, scores, thr).
//
// Reference behaviour:
//   object_detection/yolov5/utils/general.py:647-738  non_max_suppression (filter, conf = obj*cls,
//       xywh->xyxy, best-class / multi-label, class filter, top-30000, class-offset boxes, nms, max_det)
//   torchvision.ops.nms 0.11.1 (third party; called at general.py:722 and
//       object_detection/detectors/centernet/dataset.py:263): greedy, strict ">" against the double
//       threshold, fp32 areas / intersection / division, result in score-descending order.
//
// Pipeline (every stage is one launch over the whole batch; nothing returns to the host):
//   1 prefilter   wave-per-64-rows scan of the objectness column, ballot of candidates, cooperative
//                 row loads only for candidates, emits one 64-bit key per detection:
//                 [~order(score) : 32 | anchor*nc + class : 32]  -> ascending key order is
//                 "score descending, then reference candidate order", i.e. the stable refinement
//                 of the reference's unstable sorts, independent of the emission order.
//   2 sort        per-image bitonic sort of the keys (LDS resident up to 16384 keys, global+LDS beyond)
//   3 gather      rank -> box corners / conf / class rows, second key [class : 32 | rank : 32]
//   4 sort        per-image sort by (class, rank): each class becomes a contiguous, score-ordered segment
//   5 greedy      one workgroup per (class, image) segment: 64 candidates at a time are tested against
//                 the kept list (LDS, split across waves) and then resolved inside the chunk with
//                 ballot masks by one wave; boxes carry the reference's fp32 class offset so the IoUs
//                 are bit-identical; stops at max_det kept per segment
//   6 collect     ballot/prefix-scan compaction of kept ranks in rank order, first max_det rows out
// Cross-class pairs are skipped only when the image's coordinate range proves they cannot intersect
// after the 7680-px class offset; otherwise that image is handled as one segment with per-box offsets.
//
// Those six stages are the general path.  What a call usually takes (chosen by k_select, see nms_run) is shorter:
//   1'  the candidate search at validation thresholds streams the tensor (k_nms_prefilter_stream), and from the outputs of
//       the head's convolutions it walks class planes (k_nms_prefilter_conv);
//   2-6 few candidates per image: one workgroup per image does stages 2-6 in LDS (k_nms_small);
//       thousands per image, class-aware: several workgroups per image, each with a share of the classes, select the best
//       few thousand keys, sort, fetch boxes and run the greedy scans in one launch, a second launch ranks what they kept
//       (k_nms_groups, k_nms_groups_final);
//       otherwise the best k_select keys are selected and sorted in LDS (k_select_sort) ahead of stages 3-6.
// Each of them is exact for the rows it returns or flags the image (needs_full) for the next, more general attempt.
//
// Sources of the rows (nms_source): the decoded fp32 tensor; the raw head maps or the outputs of the head's convolutions,
// fp32 or fp16, decoded on the fly for the rows that need it; and, behind a half-precision model (val.py:129-130), the fp16
// decoded tensor itself or the fp16 maps with that model's roundings (half_arith) -- the kernels are instantiated per
// element type, and per arithmetic where the candidate search forms obj * cls.
#include <mutex>

#include "dh_common.h"
#include "dh_decode.h"
#include "../../include/dethead.h"

namespace {

typedef unsigned long long u64;

#define SORT_THREADS 1024
#define SORT_LDS_KEYS 16384   // 128 KiB of the CU's 160 KiB LDS: one resident sorter per CU
// Keys sorted in LDS sit at SIDX(i): one unused slot after every 16 keys.  The last pass of every bitonic stage gives a
// thread 16 consecutive keys; unpadded, the lanes of a wave are then 128 bytes apart and all fall on the same banks (a
// 32-way conflict on each of the 32 accesses); with the pad the lane stride is 17 keys and the accesses are conflict-free.
#define SIDX(i) ((i) + ((i) >> 4))
#define SORT_LDS_SLOTS(n) ((n) + (n) / 16)
#define KEY_SENTINEL 0xFFFFFFFFFFFFFFFFull
#define STAGE_KEYS 512
#define KEPT_LDS 1024

// monotone map float -> uint32 (ascending), inverted so that ascending keys mean descending scores
__device__ __forceinline__ uint32_t score_key(float s)
{
    const uint32_t b = __float_as_uint(s);
    const uint32_t e = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ~e;
}

__device__ __forceinline__ float key_score(uint32_t k)
{
    const uint32_t e = ~k;
    const uint32_t b = (e & 0x80000000u) ? (e ^ 0x80000000u) : ~e;
    return __uint_as_float(b);
}

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// Where the prediction rows come from: the decoded (bs, A, no) tensor, or (RAW) the head's raw maps, decoded on
// the fly for the rows that need it -- the same arithmetic as the decode kernel, so both sources give the same bits,
// and the decoded tensor never has to exist.
struct nms_source {
    const float* pred;
    decode_level lvl[DH_MAX_LEVELS];
    int nl, no;
    int conv_na;     // 0: lvl[].src are head maps (bs, na, ny, nx, no); na > 0: they are the outputs of the head's 1x1
                     // convolutions, (bs, na*no, ny, nx), where channel c of a row sits ny*nx elements after channel c - 1
    int corners;     // pred rows hold the corners x1, y1, x2, y2 already (k_nms_prepare_f16), not centre and size
    int half_arith;  // fp16 sources only: the roundings of a half-precision model and of NMS on its fp16 tensor (below); with
                     // nl == 0, pred is then the fp16 decoded tensor itself (a pointer in disguise, like decode_level::src)
};

// The raw maps hold fp32 or fp16 elements (T = float or _Float16; decode_level::src is then a pointer in disguise).
// fp16 values are widened to fp32 on load and everything after that is the fp32 arithmetic, so a half-precision head
// gives exactly the detections of its maps converted with .float() -- without that conversion pass.
typedef _Float16 dh_h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const _Float16* p) { return (float)*p; }
__device__ __forceinline__ float ldf_nt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ldf_nt(const _Float16* p) { return (float)__builtin_nontemporal_load(p); }
__device__ __forceinline__ dh_f4 ld4_nt(const float* p, uint32_t v)
{
    return __builtin_nontemporal_load(reinterpret_cast<const dh_f4*>(p) + v);
}
__device__ __forceinline__ dh_f4 ld4_nt(const _Float16* p, uint32_t v)
{
    const dh_h4 h = __builtin_nontemporal_load(reinterpret_cast<const dh_h4*>(p) + v);
    const dh_f4 f = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return f;
}

// channels 0..3 of row `anchor` of image b of the decoded tensor as corners (general.py:553-560); on the fp16 tensor of a
// half-precision model they are formed in fp16, like everything before the detection matrix is assembled (general.py:694)
template <typename T>
__device__ __forceinline__ float4 pred_corners(const nms_source& S, int b, int A, uint32_t anchor, int no)
{
    const T* __restrict__ p = reinterpret_cast<const T*>(S.pred) + ((size_t)b * A + anchor) * no;
    const float x = ldf(p), y = ldf(p + 1), w = ldf(p + 2), h = ldf(p + 3);
    if (S.corners) return make_float4(x, y, w, h);
    if (sizeof(T) == 2) {
        const float hw = dh_round_f16(w / 2), hh = dh_round_f16(h / 2);
        return make_float4(dh_round_f16(x - hw), dh_round_f16(y - hh), dh_round_f16(x + hw), dh_round_f16(y + hh));
    }
    return make_float4(x - w / 2, y - h / 2, x + w / 2, y + h / 2);
}

template <typename T>
struct raw_row {
    const T* p;       // first element of the row inside its map
    int l;            // level
    uint32_t local;   // row index inside the level (anchor-major, then gy, gx)
    uint32_t cs;      // distance between consecutive channels of the row (1 in a head map)
};

template <typename T>
__device__ __forceinline__ raw_row<T> raw_row_of(const nms_source& S, int b, int row)
{
    int l = 0;
#pragma unroll
    for (int k = 1; k < DH_MAX_LEVELS; ++k)
        if (k < S.nl && (uint32_t)row >= S.lvl[k].row_first) l = k;
    raw_row<T> r;
    r.l = l;
    r.local = (uint32_t)row - S.lvl[l].row_first;
    if (S.conv_na) {      // (bs, na*no, ny, nx): anchor a of the level owns `no` channel planes of ny*nx elements
        const uint32_t hw = S.lvl[l].plane.d;
        const uint32_t a = dh_div(r.local, S.lvl[l].plane);
        r.p = reinterpret_cast<const T*>(S.lvl[l].src) + (size_t)b * S.lvl[l].elems + (size_t)a * S.no * hw + (r.local - a * hw);
        r.cs = hw;
        return r;
    }
    r.p = reinterpret_cast<const T*>(S.lvl[l].src) + (size_t)b * S.lvl[l].elems + (size_t)r.local * S.no;
    r.cs = 1;
    return r;
}

// Arithmetic of the fused path on raw maps.  HALF = false: fp32 throughout (fp16 maps are widened on load).  HALF = true:
// what the reference computes behind model.half() (val.py:129-130) -- the sigmoid rounded into the fp16 tensor
// (yolo.py:58), the box of decode_box_half, conf = obj * cls rounded to fp16 (general.py:691) and the corners formed in fp16
// (general.py:694,553-560); thresholds are compared in fp32 and everything after the detection matrix is fp32, as in
// k_nms_prepare_f16, whose results on the decoded fp16 tensor this reproduces bit for bit.
template <bool HALF> __device__ __forceinline__ float act_of(float x)
{
    const float s = dh_sigmoid(x);                                   // yolo.py:58,62
    return HALF ? dh_round_f16(s) : s;
}
template <bool HALF> __device__ __forceinline__ float conf_of(float cls, float obj)
{
    const float p = cls * obj;                                       // general.py:691 (exact in fp32 for two fp16 factors)
    return HALF ? dh_round_f16(p) : p;
}

// corners of raw row `anchor` of image b (general.py:553-560 on the decoded row)
template <typename T>
__device__ __forceinline__ float4 raw_corners(const nms_source& S, int b, uint32_t anchor)
{
    const raw_row<T> rr = raw_row_of<T>(S, b, (int)anchor);
    const decode_level& L = S.lvl[rr.l];
    const float sx = dh_sigmoid(ldf(rr.p)), sy = dh_sigmoid(ldf(rr.p + rr.cs));
    const float sw = dh_sigmoid(ldf(rr.p + 2 * (size_t)rr.cs)), sh = dh_sigmoid(ldf(rr.p + 3 * (size_t)rr.cs));
    if (sizeof(T) == 2 && S.half_arith) {
        const float x = decode_box_half(L, rr.local, 0, dh_round_f16(sx)), y = decode_box_half(L, rr.local, 1, dh_round_f16(sy));
        const float hw = dh_round_f16(decode_box_half(L, rr.local, 2, dh_round_f16(sw)) / 2);
        const float hh = dh_round_f16(decode_box_half(L, rr.local, 3, dh_round_f16(sh)) / 2);
        return make_float4(dh_round_f16(x - hw), dh_round_f16(y - hh), dh_round_f16(x + hw), dh_round_f16(y + hh));
    }
    const float x = decode_box(L, rr.local, 0, sx), y = decode_box(L, rr.local, 1, sy);
    const float w = decode_box(L, rr.local, 2, sw), h = decode_box(L, rr.local, 3, sh);
    return make_float4(x - w / 2, y - h / 2, x + w / 2, y + h / 2);
}

// ------------------------------------------------------------------------------------------ 1 prefilter
struct row_best {
    float v;
    int c;
};

__device__ __forceinline__ row_best wave_argmax_first(float v, int c)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oc = __shfl_xor(c, o, 64);
        if (ov > v || (ov == v && oc < c)) { v = ov; c = oc; }
    }
    row_best r = {v, c};
    return r;
}

// the lanes of one wave have all passed this point, and their LDS accesses before it are visible to each other
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#define DENSE_ROWS 16        // rows per LDS tile of the dense single-label path
#define DENSE_REGS 24        // 64 * DENSE_REGS >= DENSE_ROWS * no, i.e. no <= 96
#define DENSE_MIN 24         // candidates among a wave's 64 rows from which the dense path is taken
#define DENSE_VREGS (DENSE_REGS / 4)   // 16-byte vectors per lane and tile on the vector form of that path

// NCH > 0: the class scores of the rows in flight are fetched up front as NCH chunks of 64 classes
// (nc <= 64*NCH), which keeps 4*NCH independent loads in flight per wave; NCH = 0 handles any nc
// chunk by chunk.
template <bool MULTI, int NCH, bool RAW, typename T = float, bool HALF = false>
__global__ void __launch_bounds__(256) k_nms_prefilter(const nms_source S, const float* __restrict__ obj_plane,
                                                       int A, int nc, float conf,
                                                       const unsigned char* __restrict__ class_keep, int cap,
                                                       u64* __restrict__ keys, int cap_pad, int* __restrict__ counts,
                                                       int dense)
{
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int no = nc + 5;
    const T* __restrict__ img = RAW ? nullptr : reinterpret_cast<const T*>(S.pred) + (size_t)b * A * no;
    u64* __restrict__ out = keys + (size_t)b * cap_pad;
    __shared__ u64 stage[4][STAGE_KEYS];
    extern __shared__ __align__(16) float s_dense[];   // single-label, dense != 0: 4 waves x DENSE_ROWS rows x no floats
    int pending = 0;

    // single label: the workgroup gathers its keys in LDS (all of `stage`) and takes room in the image's key list with one
    // atomic per ~1800 keys -- one atomic per wave and 64 rows, all on the image's one counter, serialise in L2
    // (measured: 1563 of them per image cost 0.6 ms on a dense 100 000-row tensor)
    u64* const bstage = &stage[0][0];
    __shared__ int s_cnt[4], s_base;
    int bpending = 0;
    for (int base0 = blockIdx.x * 256; base0 < A; base0 += gridDim.x * 256) {      // workgroup-uniform trip count
        const int row0 = base0 + wave * 64;
        const int row = row0 + lane;
        float obj = 0.0f;
        if (row < A) {
            // the strided column is touched once: non-temporal loads fetch single 64-byte sectors (2.2x faster cold)
            if (RAW) obj = act_of<HALF>(ldf_nt(raw_row_of<T>(S, b, row).p + 4));
            else obj = obj_plane ? obj_plane[(size_t)b * A + row] : ldf_nt(img + (size_t)row * no + 4);
        }
        const bool cand = (row < A) && (obj > conf);                       // general.py:656,675
        u64 todo = __ballot(cand);
        float my_best = 0.0f;
        int my_cls = 0;
        // Single label, most of the wave's 64 rows are candidates (a dense prediction tensor or a low threshold): the
        // rows are consecutive in memory, so they are brought into LDS 16 at a time with coalesced loads -- every load
        // of a tile is issued before the first use -- and four lanes per row find the best class there.  Fetching
        // candidate rows one group of four at a time costs a memory round trip per group (measured 1.7 TB/s).
        if (!MULTI && dense && __popcll(todo) >= DENSE_MIN &&
            (!RAW || raw_row_of<T>(S, b, row0).l == raw_row_of<T>(S, b, min(row0 + 63, A - 1)).l)) {
            float* __restrict__ tile = s_dense + wave * (DENSE_ROWS * no);
            // dense == 2 (decoded fp32 tensor, 16-byte aligned, A * no % 4 == 0): the full tiles of the wave's 64 rows come in
            // as 16-byte vectors, DENSE_ROWS * no / 4 of them per tile, and the next tile's loads are issued -- unconditionally,
            // so that waiting for this tile's vectors does not mean waiting for those -- before this tile is looked at.
            int sub_first = 0;
            if (!RAW && sizeof(T) == 4 && dense == 2) {
                const int full = min(64 / DENSE_ROWS, (A - row0) / DENSE_ROWS);     // wave-uniform
                const int nv = DENSE_ROWS / 4 * no;                                  // vectors per tile
                const dh_f4* __restrict__ rows4 = reinterpret_cast<const dh_f4*>(reinterpret_cast<const float*>(S.pred) + ((size_t)b * A + row0) * no);
                dh_f4 buf[2][DENSE_VREGS];
                if (full > 0) {
#pragma unroll
                    for (int i = 0; i < DENSE_VREGS; ++i) buf[0][i] = rows4[min(i * 64 + lane, nv - 1)];
                }
#pragma unroll
                for (int sub = 0; sub < 64 / DENSE_ROWS; ++sub) {
                    if (sub >= full) break;
                    const int nx = sub + 1 < full ? sub + 1 : sub;                   // no next tile: its own once more
#pragma unroll
                    for (int i = 0; i < DENSE_VREGS; ++i) buf[(sub + 1) & 1][i] = rows4[(size_t)nx * nv + min(i * 64 + lane, nv - 1)];
#pragma unroll
                    for (int i = 0; i < DENSE_VREGS; ++i)
                        if (i * 64 + lane < nv) reinterpret_cast<dh_f4*>(tile)[i * 64 + lane] = buf[sub & 1][i];
                    wave_sync();
                    const int rr = lane >> 2, part = lane & 3;
                    const float myo = __shfl(obj, DENSE_ROWS * sub + rr, 64);
                    float qv = -INFINITY;
                    int qc = 0x7fffffff;
                    const float* __restrict__ rowp = tile + rr * no + 5;
#pragma unroll 4
                    for (int c = part; c < nc; c += 4) {
                        const float sc = conf_of<HALF>(rowp[c], myo);                           // general.py:691
                        if (sc > qv) { qv = sc; qc = c; }
                    }
#pragma unroll
                    for (int d = 2; d > 0; d >>= 1) {
                        const float ov = __shfl_xor(qv, d, 64);
                        const int oc = __shfl_xor(qc, d, 64);
                        if (ov > qv || (ov == qv && oc < qc)) { qv = ov; qc = oc; }
                    }
                    const float rv = __shfl(qv, 4 * (lane & (DENSE_ROWS - 1)), 64);             // general.py:701 (first max)
                    const int rc = __shfl(qc, 4 * (lane & (DENSE_ROWS - 1)), 64);
                    if ((lane / DENSE_ROWS) == sub) { my_best = rv; my_cls = rc; }
                    wave_sync();
                }
                sub_first = full > 0 ? full : 0;
            }
#pragma unroll 1
            for (int sub = sub_first; sub < 64 / DENSE_ROWS; ++sub) {
                const int rs = row0 + DENSE_ROWS * sub;
                if (rs >= A) break;
                if (((todo >> (DENSE_ROWS * sub)) & ((1ull << DENSE_ROWS) - 1ull)) == 0) continue;
                const int n_el = min(DENSE_ROWS, A - rs) * no;
                const T* __restrict__ base = RAW ? raw_row_of<T>(S, b, rs).p : img + (size_t)rs * no;
                float reg[DENSE_REGS];
#pragma unroll
                for (int i = 0; i < DENSE_REGS; ++i) reg[i] = (i * 64 + lane < n_el) ? ldf(base + i * 64 + lane) : 0.0f;
#pragma unroll
                for (int i = 0; i < DENSE_REGS; ++i)
                    if (i * 64 + lane < n_el) tile[i * 64 + lane] = reg[i];
                wave_sync();
                const int rr = lane >> 2, part = lane & 3;
                const float myo = __shfl(obj, DENSE_ROWS * sub + rr, 64);
                float qv = -INFINITY;
                int qc = 0x7fffffff;
                if (rr * no < n_el) {
                    const float* __restrict__ rowp = tile + rr * no + 5;
                    for (int c = part; c < nc; c += 4) {
                        const float sc = conf_of<HALF>(RAW ? act_of<HALF>(rowp[c]) : rowp[c], myo);   // yolo.py:62, general.py:691
                        if (sc > qv) { qv = sc; qc = c; }
                    }
                }
#pragma unroll
                for (int d = 2; d > 0; d >>= 1) {
                    const float ov = __shfl_xor(qv, d, 64);
                    const int oc = __shfl_xor(qc, d, 64);
                    if (ov > qv || (ov == qv && oc < qc)) { qv = ov; qc = oc; }
                }
                const float rv = __shfl(qv, 4 * (lane & (DENSE_ROWS - 1)), 64);                 // general.py:701 (first max)
                const int rc = __shfl(qc, 4 * (lane & (DENSE_ROWS - 1)), 64);
                if ((lane / DENSE_ROWS) == sub) { my_best = rv; my_cls = rc; }
                wave_sync();
            }
            todo = 0;
        }
        while (todo) {
            // up to 4 candidate rows in flight per wave
            int r[4];
            float o[4];
            int nr = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (todo) {
                    r[k] = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    nr = k + 1;
                } else {
                    r[k] = r[0];
                }
                o[k] = __shfl(obj, r[k], 64);
            }
            if (!MULTI) {
                // best class of up to four rows at once: quarter q of the wave (16 lanes) owns row r[q], every lane scans
                // the classes j, j+16, ... of its row (first maximum: ascending classes, strict >), then four exchange
                // steps inside the quarter instead of six across the wave per row
                const int q = lane >> 4, j = lane & 15;
                const int myr = q == 0 ? r[0] : (q == 1 ? r[1] : (q == 2 ? r[2] : r[3]));
                const float myo = q == 0 ? o[0] : (q == 1 ? o[1] : (q == 2 ? o[2] : o[3]));
                const T* qp = (RAW ? raw_row_of<T>(S, b, row0 + myr).p
                                   : img + (size_t)(row0 + myr) * no) + 5;
                float qv = -INFINITY;
                int qc = 0x7fffffff;
                if (NCH > 0) {
                    float pre16[NCH > 0 ? NCH * 4 : 1];
#pragma unroll
                    for (int h = 0; h < NCH * 4; ++h) pre16[h] = (h * 16 + j < nc) ? ldf(qp + h * 16 + j) : 0.0f;
#pragma unroll
                    for (int h = 0; h < NCH * 4; ++h) {
                        const int c = h * 16 + j;
                        const float sc = conf_of<HALF>(RAW ? act_of<HALF>(pre16[h]) : pre16[h], myo);   // yolo.py:62, general.py:691
                        if (c < nc && sc > qv) { qv = sc; qc = c; }
                    }
                } else {
                    for (int c = j; c < nc; c += 16) {
                        const float x = ldf(qp + c);
                        const float sc = conf_of<HALF>(RAW ? act_of<HALF>(x) : x, myo);
                        if (sc > qv) { qv = sc; qc = c; }
                    }
                }
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) {
                    const float ov = __shfl_xor(qv, d, 64);
                    const int oc = __shfl_xor(qc, d, 64);
                    if (ov > qv || (ov == qv && oc < qc)) { qv = ov; qc = oc; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float rv = __shfl(qv, k * 16, 64);                                    // general.py:701 (first max)
                    const int rc = __shfl(qc, k * 16, 64);
                    if (k < nr && lane == r[k]) { my_best = rv; my_cls = rc; }
                }
                continue;
            }
            float bv[4];
            int bc[4];
            const T* rp[4];          // first class score of each row in flight (wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bv[k] = -INFINITY;
                bc[k] = 0x7fffffff;
                rp[k] = (RAW ? raw_row_of<T>(S, b, row0 + r[k]).p : img + (size_t)(row0 + r[k]) * no) + 5;
            }
            float pre[4][NCH > 0 ? NCH : 1];
            if (NCH > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int h = 0; h < NCH; ++h)
                        pre[k][h] = (h * 64 + lane < nc) ? ldf(rp[k] + h * 64 + lane) : 0.0f;
            }
            const int nch = NCH > 0 ? NCH : (nc + 63) / 64;
#pragma unroll
            for (int h = 0; h < nch; ++h) {
                const int c = h * 64 + lane;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (NCH > 0) v[k] = pre[k][NCH > 0 ? h : 0];
                    else v[k] = (c < nc) ? ldf(rp[k] + c) : 0.0f;
                    if (RAW) v[k] = act_of<HALF>(v[k]);        // yolo.py:62
                }
                const bool keepc = (c < nc) && (class_keep == nullptr || class_keep[c] != 0);   // general.py:705-706
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s = conf_of<HALF>(v[k], o[k]);                                 // general.py:691
                    if (MULTI) {
                        const bool pass = (k < nr) && keepc && (s > conf);                     // general.py:698
                        const u64 pm = __ballot(pass);
                        if (pm) {
                            if (pass) {
                                const int slot = pending + __popcll(pm & lanemask_lt());
                                stage[wave][slot] = ((u64)score_key(s) << 32) | (uint32_t)((row0 + r[k]) * nc + c);
                            }
                            pending += __popcll(pm);
                        }
                    } else if (c < nc && s > bv[k]) {
                        bv[k] = s;
                        bc[k] = c;
                    }
                }
                if (MULTI && pending > STAGE_KEYS - 4 * 64) {   // flush: one atomic per >256 keys; 256 more always fit
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&counts[b], pending);
                    base = __shfl(base, 0, 64);
                    for (int i = lane; i < pending; i += 64)
                        if (base + i < cap) out[base + i] = stage[wave][i];
                    pending = 0;
                }
            }
            if (!MULTI) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const row_best rb = wave_argmax_first(bv[k], bc[k]);                       // general.py:701 (first max)
                    if (k < nr && lane == r[k]) { my_best = rb.v; my_cls = rb.c; }
                }
            }
        }
        if (!MULTI) {
            const bool pass = cand && (my_best > conf) && (class_keep == nullptr || class_keep[my_cls] != 0);
            const u64 pm = __ballot(pass);
            if (lane == 0) s_cnt[wave] = __popcll(pm);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = s_cnt[w]; if (w < wave) before += c; total += c; }
            if (pass) bstage[bpending + before + __popcll(pm & lanemask_lt())] =
                ((u64)score_key(my_best) << 32) | (uint32_t)(row * nc + my_cls);
            bpending += total;
            __syncthreads();
            if (bpending > 4 * STAGE_KEYS - 256) {              // the next round adds at most 256
                if (threadIdx.x == 0) s_base = atomicAdd(&counts[b], bpending);
                __syncthreads();
                for (int i = threadIdx.x; i < bpending; i += 256)
                    if (s_base + i < cap) out[s_base + i] = bstage[i];
                bpending = 0;
                __syncthreads();
            }
        }
    }
    if (!MULTI && bpending) {                                   // workgroup-uniform
        if (threadIdx.x == 0) s_base = atomicAdd(&counts[b], bpending);
        __syncthreads();
        for (int i = threadIdx.x; i < bpending; i += 256)
            if (s_base + i < cap) out[s_base + i] = bstage[i];
    }
    if (MULTI) {                                                // what the four waves still hold goes out with one atomic
        if (lane == 0) s_cnt[wave] = pending;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = s_cnt[w]; if (w < wave) before += c; total += c; }
        if (total) {
            if (threadIdx.x == 0) s_base = atomicAdd(&counts[b], total);
            __syncthreads();
            const int base = s_base + before;
            for (int i = lane; i < pending; i += 64)
                if (base + i < cap) out[base + i] = stage[wave][i];
        }
    }
}

// ------------------------------------------------------------------------------------------ 1'' prefilter, convolution layout
// The candidates straight from the outputs of the head's 1x1 convolutions, (bs, na*no, ny, nx): there the objectness
// logits of one (image, anchor) are a contiguous plane and so is every class, so nothing has to be decomposed into
// (row, channel).  A thread owns one position and reads its objectness with a coalesced load.  What follows depends on how
// many of a wave's 64 positions are candidates:
//   many (validation thresholds): position-parallel -- every lane walks the class planes of its own position, 16 coalesced
//         loads in flight per lane, so the whole tensor is streamed once at memory speed;
//   few (inference thresholds):   class-parallel -- the wave takes its candidates four at a time, lane = class, one
//         round trip per group instead of a walk over all planes for the sake of one or two lanes.
// Keys are gathered in LDS per workgroup and reach the image's key list with one atomic per workgroup (a wave whose
// keys do not fit the stage any more writes them directly).  Same key set as k_nms_prefilter on the decoded tensor:
// sigmoid and the product obj * cls are the arithmetic of the decode followed by general.py:691.
#define CONV_CH 16              // class planes in flight per lane on the position-parallel path
#define CONV_DENSE 5            // candidates in a wave from which that path is taken
#define CONV_STAGE 2048         // keys a workgroup gathers in LDS

struct conv_blocks {
    uint32_t blocks[DH_MAX_LEVELS];    // na * ceil(ny*nx / 256) per level
};

// one key per passing lane: into the workgroup's stage, or straight into the image's list when the stage is full.
// Called by whole waves.  s_n only grows: a reservation that does not fit the stage is never handed back (a later,
// smaller one could otherwise land beyond keys that were never written), it lowers s_valid to where it started, and
// since every reservation after it starts beyond the capacity as well, the staged keys are exactly [0, s_valid).
__device__ __forceinline__ void conv_emit(bool pass, u64 key, u64* stage, int* s_n, int* s_valid, u64* __restrict__ out, int* counter,
                                          int cap)
{
    const u64 pm = __ballot(pass);
    if (!pm) return;
    const int cnt = __popcll(pm), lane = threadIdx.x & 63;
    int first = 0, direct = 0;
    if (lane == 0) {
        first = atomicAdd(s_n, cnt);
        if (first + cnt > CONV_STAGE) {
            atomicMin(s_valid, first);
            first = atomicAdd(counter, cnt);
            direct = 1;
        }
    }
    first = __shfl(first, 0, 64);
    direct = __shfl(direct, 0, 64);
    if (!pass) return;
    const int at = first + __popcll(pm & lanemask_lt());
    if (!direct) stage[at] = key;
    else if (at < cap) out[at] = key;
}

template <bool MULTI, typename T, bool HALF = false>
__global__ void __launch_bounds__(256) k_nms_prefilter_conv(const nms_source S, const conv_blocks CB, int nc, float conf,
                                                            const unsigned char* __restrict__ class_keep, int cap,
                                                            u64* __restrict__ keys, int cap_pad, int* __restrict__ counts)
{
    __shared__ u64 stage[CONV_STAGE];
    __shared__ int s_n, s_valid, s_base;
    const int b = blockIdx.y;
    uint32_t blk = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int k = 0; k < DH_MAX_LEVELS - 1; ++k)
        if (l == k && k + 1 < S.nl && blk >= CB.blocks[k]) { blk -= CB.blocks[k]; l = k + 1; }
    const decode_level& L = S.lvl[l];
    const uint32_t hw = L.plane.d;
    const uint32_t tiles = (hw + 255) / 256;
    const uint32_t a = blk / tiles;
    const int lane = threadIdx.x & 63;
    const uint32_t s_wave = (blk - a * tiles) * 256 + (threadIdx.x & ~63u);     // first position of this wave
    const uint32_t s = s_wave + lane;
    // channel c of position q of this (image, anchor) sits at plane0[c * hw + q]
    const T* __restrict__ plane0 = reinterpret_cast<const T*>(L.src) + (size_t)b * L.elems + (size_t)a * S.no * hw;
    const uint32_t row_wave = L.row_first + a * hw + s_wave;          // row of the decoded tensor (yolo.py:68,70)
    u64* __restrict__ out = keys + (size_t)b * cap_pad;
    if (threadIdx.x == 0) { s_n = 0; s_valid = CONV_STAGE; }
    __syncthreads();
    float obj = 0.0f;
    if (s < hw) obj = act_of<HALF>(ldf_nt(plane0 + 4 * (size_t)hw + s));
    const bool cand = s < hw && obj > conf;                           // general.py:656,675
    u64 todo = __ballot(cand);
    float best = -INFINITY;
    int best_c = 0;
    if (__popcll(todo) >= CONV_DENSE) {
        // Multi-label: sigmoid(x) * obj > conf needs x > logit(conf / obj), so one compare against a per-position bound
        // (lowered by far more than the rounding of either side can amount to) rules out almost every class without a
        // sigmoid; the exact test runs only where some lane of the wave gets past it.
        float bound = -INFINITY;
        if (MULTI && cand) {
            // < 1 for a candidate.  With the fp16 roundings the sigmoid and the product can each round up by 2^-11 of their
            // value: the bound is taken for a threshold lowered by more than both (and not at all below 2^-13, where fp16
            // values are spaced by a fixed amount)
            const float q = (HALF ? 0.998f : 1.0f) * conf / obj;
            if (q > 0.0f && 1.0f - q > 1e-3f && !(HALF && conf < 1.3e-4f)) bound = __logf(q / (1.0f - q)) - 0.01f;
        }
        // The class planes are walked CONV_CH at a time, the next group's loads in flight while this one is looked at.  Every
        // lane loads, candidate or not, and a group past the last class re-reads the last plane: loads behind a condition
        // would make the join wait for the newest load in flight instead of the oldest.
        const uint32_t sq = min(s, hw - 1);
        auto fetch = [&](float (&v)[CONV_CH], int c0) {
#pragma unroll
            for (int u = 0; u < CONV_CH; ++u)      // element offsets inside one (image, anchor) stay below 2^31 (checked on the host)
                v[u] = ldf_nt(plane0 + ((uint32_t)(5 + min(c0 + u, nc - 1)) * hw + sq));
        };
        auto take = [&](const float (&v)[CONV_CH], int c0) {
#pragma unroll
            for (int u = 0; u < CONV_CH; ++u) {
                const int c = c0 + u;
                if (c >= nc) break;                                   // wave-uniform
                if (MULTI) {
                    if (__ballot(cand && v[u] > bound) == 0) continue;
                    if (class_keep != nullptr && class_keep[c] == 0) continue;                              // general.py:705
                    const float sc = conf_of<HALF>(act_of<HALF>(v[u]), obj);   // yolo.py:62, general.py:691
                    conv_emit(cand && sc > conf, ((u64)score_key(sc) << 32) | (uint32_t)((row_wave + lane) * nc + c), stage,
                              &s_n, &s_valid, out, &counts[b], cap);  // general.py:698
                } else {
                    const float sc = conf_of<HALF>(act_of<HALF>(v[u]), obj);
                    if (cand && sc > best) {                          // general.py:701 (first maximum)
                        best = sc;
                        best_c = c;
                    }
                }
            }
        };
        float va[CONV_CH], vb[CONV_CH];
        fetch(va, 0);
#pragma unroll 1
        for (int c0 = 0; c0 < nc; c0 += 2 * CONV_CH) {
            fetch(vb, c0 + CONV_CH);
            take(va, c0);
            fetch(va, c0 + 2 * CONV_CH);
            take(vb, c0 + CONV_CH);
        }
    } else {
        while (todo) {
            int r[4];
            float o[4];
            int nr = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (todo) {
                    r[k] = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    nr = k + 1;
                } else {
                    r[k] = r[0];
                }
                o[k] = __shfl(obj, r[k], 64);
            }
            float bv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int bc[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
            for (int h0 = 0; h0 < nc; h0 += 128) {                    // two chunks of 64 classes per round trip
                float x[4][2];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int c = h0 + 64 * h + lane;
                        x[k][h] = c < nc ? ldf(plane0 + (size_t)(5 + c) * hw + s_wave + r[k]) : 0.0f;
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = h0 + 64 * h + lane;
                    if (h0 + 64 * h >= nc) break;                     // wave-uniform
                    const bool keepc = c < nc && (class_keep == nullptr || class_keep[c] != 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sc = conf_of<HALF>(act_of<HALF>(x[k][h]), o[k]);  // yolo.py:62, general.py:691
                        if (MULTI) {
                            conv_emit(k < nr && keepc && sc > conf,
                                      ((u64)score_key(sc) << 32) | (uint32_t)((row_wave + r[k]) * nc + c), stage, &s_n,
                                      &s_valid, out, &counts[b], cap);
                        } else if (c < nc && sc > bv[k]) {
                            bv[k] = sc;
                            bc[k] = c;
                        }
                    }
                }
            }
            if (!MULTI) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const row_best rb = wave_argmax_first(bv[k], bc[k]);      // general.py:701 (first maximum)
                    if (k < nr && lane == r[k]) { best = rb.v; best_c = rb.c; }
                }
            }
        }
    }
    if (!MULTI) {
        const bool pass = cand && best > conf && (class_keep == nullptr || class_keep[best_c] != 0);   // general.py:702,705
        conv_emit(pass, ((u64)score_key(best) << 32) | (uint32_t)((row_wave + lane) * nc + best_c), stage, &s_n, &s_valid, out,
                  &counts[b], cap);
    }
    __syncthreads();
    const int n = min(s_n, s_valid);                                  // the staged keys (see conv_emit)
    if (n == 0) return;                                               // workgroup-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(&counts[b], n);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if (s_base + i < cap) out[s_base + i] = stage[i];
}

// ------------------------------------------------------------------------------------------ 1' prefilter, streaming
// Multi-label candidates are (row, class) pairs that are judged one by one (obj > conf and obj*cls > conf,
// general.py:656,691,698): no per-row reduction is needed, so when most rows are candidates anyway (validation
// thresholds) the whole tensor is streamed once with coalesced 16-byte loads -- at several times the rate of fetching
// candidate rows one by one -- and every element with a class channel is tested where it lands.  The key set is the
// same as k_nms_prefilter<true>'s; only the emission order differs, and the sort that follows does not depend on it.
// The kernel is bound by instruction issue, not by memory, so the work per 16-byte vector is kept short:
//   * a vector's four elements lie in one row, or run over into the box channels of the next one, which are never class
//     scores: one row index and one objectness value serve the whole vector;
//   * the streaming loop only asks whether any element of the vector could pass -- the largest of the four against the
//     threshold, with straight-line code -- and sets the few vectors that could (a few in a hundred) aside in a queue the
//     wave keeps in LDS; vectors holding box channels of a candidate row are set aside more often than needed;
//   * the wave tests what it has set aside 64 vectors at a time, all lanes busy: exact arithmetic per element, class
//     filter, keys.  The divergent code of the loop is the dozen instructions that append to the queue;
//   * keys go to one LDS stage per workgroup and reach the image's key list with one global atomic per workgroup.  That
//     atomic and the copy behind it are two round trips through a memory system whose queues the streaming loads keep full
//     (several microseconds each), with the workgroup idle meanwhile: so the grid is sized to what the chip holds at once,
//     and every workgroup streams many tiles and pays the two round trips once, at the very end.
#define STREAM_U 4           // vectors per lane and tile: their loads are all issued before the first use
#define STREAM_TILE (256 * STREAM_U)   // vectors per tile
#define STREAM_MIN_TILES 8   // a workgroup streams at least this many tiles (if its image has them)
#define STREAM_WG_PER_CU 4   // workgroups a compute unit holds (registers: 4 waves per SIMD)
#define STREAM_ROUNDS 2       // the grid is this many times what the chip holds at once
#define STREAM_QUEUE 256     // vectors a wave can set aside; tested whenever 64 or more are waiting
#define STREAM_STAGE 1984    // keys a workgroup gathers in LDS (with the queues: just under 40 KB, four workgroups per
                             // compute unit); further ones go out wave by wave
#define STREAM_SEG_SHIFT 29  // a queued vector's index (< 2^29: an image's segment has < 2^31 elements) and, above it, its segment

struct stream_seg {
    const float* src;      // (bs, elems): one level's raw map (fp32 or fp16 elements), or the whole decoded tensor
    uint32_t elems;        // elements per image
    uint32_t row_first;    // first row (anchor) of this segment inside an image
    uint32_t tiles;        // ceil(elems / 4 / STREAM_TILE)
};

struct stream_params {
    stream_seg seg[DH_MAX_LEVELS];
    int nseg;
    uint32_t tiles;            // per image, all segments
    dh_fastdiv no;
};

// Which of a vector's four elements are candidates, bit k for element k, and their scores: channel c0 + k of a row whose
// objectness o is above the threshold.
template <bool RAW, bool HALF>
__device__ __forceinline__ uint32_t stream_exact(const dh_f4 x, float o, uint32_t c0, uint32_t no, float conf,
                                                 const unsigned char* __restrict__ class_keep, float sc[4])
{
    const float xs[4] = {x.x, x.y, x.z, x.w};
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t c = c0 + k - 5u;                              // class, if the channel is in [5, no)
        sc[k] = conf_of<HALF>(RAW ? act_of<HALF>(xs[k]) : xs[k], o);       // yolo.py:62, general.py:691
        bool ok = c < no - 5u && sc[k] > conf;                       // general.py:698
        if (class_keep != nullptr && ok) ok = class_keep[c] != 0;    // general.py:705-706
        bits |= ok ? 1u << k : 0u;
    }
    return bits;
}

// One lane's vector straight to the image's key list, for the rare lane that finds its wave's queue full.
template <bool RAW, bool HALF>
__device__ __attribute__((noinline)) void stream_direct(const dh_f4 x, float o, uint32_t v, uint32_t c0, const dh_fastdiv no,
                                                        float conf, const unsigned char* __restrict__ class_keep,
                                                        uint32_t row_first, int nc, int* __restrict__ count, int cap,
                                                        u64* __restrict__ out)
{
    float sc[4];
    const uint32_t row = dh_div(v * 4, no);
    const uint32_t bits = stream_exact<RAW, HALF>(x, o, c0, no.d, conf, class_keep, sc);
    if (!bits) return;
    int at = atomicAdd(count, __popc(bits));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (bits >> k & 1) {
            if ((uint32_t)at < (uint32_t)cap) out[at] = ((u64)score_key(sc[k]) << 32) | (uint32_t)((row_first + row) * nc + (c0 + k - 5));
            ++at;
        }
}

template <bool RAW, typename T = float, bool HALF = false>
__global__ void __launch_bounds__(256, STREAM_WG_PER_CU) k_nms_prefilter_stream(const stream_params P, const float* __restrict__ obj_plane,
                                                              int A, int nc, float conf,
                                                              const unsigned char* __restrict__ class_keep, int cap,
                                                              u64* __restrict__ keys, int cap_pad, int* __restrict__ counts)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // tiles t0, t0 + step, ...: dense and sparse stretches of the image (the coarse level's rows hold several times the
    // candidates of the fine level's) are dealt out evenly, so that no workgroup gathers many more keys than its stage holds
    const uint32_t t0 = blockIdx.x, step = gridDim.x, t1 = P.tiles;
    if (t0 >= t1) return;                                            // workgroup-uniform, before any barrier
    const uint32_t no = P.no.d;
    u64* __restrict__ out = keys + (size_t)b * cap_pad;
    __shared__ dh_f4 qx[4][STREAM_QUEUE];          // the vectors set aside, per wave,
    __shared__ uint2 qm[4][STREAM_QUEUE];          // with their row's objectness and their index
    __shared__ u64 stage[STREAM_STAGE];
    __shared__ uint32_t s_row_first[DH_MAX_LEVELS];
    __shared__ int q_n[4], s_n, s_valid, s_base;
    if (threadIdx.x < 4) q_n[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_n = 0; s_valid = STREAM_STAGE;
#pragma unroll
        for (int k = 0; k < DH_MAX_LEVELS; ++k) s_row_first[k] = k < P.nseg ? P.seg[k].row_first : 0;
    }
    __syncthreads();
    // raw maps: the largest element goes through the sigmoid alone; the margin covers any non-monotonic rounding, and with
    // the fp16 roundings also a product that rounds up to the threshold (at most 2^-11 of its value)
    // (and below 2^-14, where fp16 is spaced 2^-24 apart, by half that spacing)
    const float conf_pre = HALF ? conf * 0.998f - 6.0e-8f : (RAW ? conf * 0.9999f : conf);
    // decoded tensor (one segment): a row's objectness is in the compact plane the decode left, or in channel 4 of the row
    const float* __restrict__ plane = (!RAW && obj_plane) ? obj_plane + (size_t)b * A + P.seg[0].row_first : nullptr;
    const T* __restrict__ column = reinterpret_cast<const T*>(P.seg[0].src) + (size_t)b * P.seg[0].elems + 4;
    // tests the n vectors the wave has set aside (all 64 lanes call this together)
    auto drain = [&](int n) {
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool live = i < n;
            const dh_f4 xv = qx[w][live ? i : 0];
            const uint2 m = qm[w][live ? i : 0];
            const float o = __uint_as_float(m.x);
            const uint32_t e = (m.y & ((1u << STREAM_SEG_SHIFT) - 1)) * 4, row = dh_div(e, P.no), c0 = e - row * no;
            const uint32_t row_g = s_row_first[m.y >> STREAM_SEG_SHIFT] + row;
            float sc[4];
            uint32_t bits = stream_exact<RAW, HALF>(xv, o, c0, no, conf, class_keep, sc);
            if (!live) bits = 0;
            const int cnt = __popc(bits);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            const int total = __shfl(incl, 63);
            if (total == 0) continue;                                // wave-uniform
            int first = 0, direct = 0;
            if (lane == 0) {
                first = atomicAdd(&s_n, total);
                direct = first + total > STREAM_STAGE;               // no room in the stage: straight to the image's list
                if (direct) {                                        // (s_n only grows, see conv_emit)
                    atomicMin(&s_valid, first);
                    first = atomicAdd(&counts[b], total);
                }
            }
            first = __shfl(first, 0);
            direct = __shfl(direct, 0);
            int at = first + incl - cnt;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (bits >> k & 1) {
                    const u64 key = ((u64)score_key(sc[k]) << 32) | (uint32_t)(row_g * nc + (c0 + k - 5));
                    if (!direct) stage[at] = key;
                    else if ((uint32_t)at < (uint32_t)cap) out[at] = key;
                    ++at;
                }
        }
    };
    // one tile's loads, all issued before anything waits for them; vi = vector index in its segment | segment << 29
    auto issue = [&](uint32_t t, dh_f4 (&x)[STREAM_U], float (&oraw)[STREAM_U], uint32_t (&c0)[STREAM_U], uint32_t (&vi)[STREAM_U]) {
        const bool tile_live = t < t1;                               // a tile past the end is asked for, and ignored
        uint32_t tt = tile_live ? t : t0;
        int l = 0;
#pragma unroll
        for (int k = 0; k < DH_MAX_LEVELS - 1; ++k)
            if (l == k && k + 1 < P.nseg && tt >= P.seg[k].tiles) { tt -= P.seg[k].tiles; l = k + 1; }
        const stream_seg& G = P.seg[l];
        const uint32_t nvec = G.elems >> 2;
        const T* __restrict__ src = reinterpret_cast<const T*>(G.src) + (size_t)b * G.elems;
        const uint32_t v0 = tt * STREAM_TILE + threadIdx.x;
#pragma unroll
        for (int u = 0; u < STREAM_U; ++u) {
            const uint32_t v = v0 + u * 256;
            const bool live = tile_live && v < nvec;
            const uint32_t vv = live ? v : 0;
            vi[u] = vv | ((uint32_t)l << STREAM_SEG_SHIFT);
            const uint32_t e = vv * 4, row = dh_div(e, P.no);
            c0[u] = live ? e - row * no : no;                       // c0 = no: a dead vector has no class channel
            x[u] = ld4_nt(src, vv);
            if constexpr (RAW) oraw[u] = ldf(src + (size_t)row * no + 4);
            else oraw[u] = plane ? plane[row] : ldf(column + (size_t)row * no);
        }
    };
    // one tile's vectors: set aside those that may hold a candidate
    auto sift = [&](const dh_f4 (&x)[STREAM_U], const float (&oraw)[STREAM_U], const uint32_t (&c0)[STREAM_U],
                    const uint32_t (&vi)[STREAM_U]) {
#pragma unroll
        for (int u = 0; u < STREAM_U; ++u) {
            const float o = RAW ? act_of<HALF>(oraw[u]) : oraw[u];
            const bool has_cls = c0[u] - 2u < no - 2u;              // some channel c0 .. c0+3 lies in [5, no)
            float mx;
            if (RAW) {
                mx = act_of<HALF>(fmaxf(fmaxf(x[u].x, x[u].y), fmaxf(x[u].z, x[u].w)));
            } else {                                                // decoded box channels are pixels, far above any score:
                const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w};       // only class channels enter the maximum
                mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < 4; ++k) mx = fmaxf(mx, c0[u] + k - 5u < no - 5u ? xs[k] : -INFINITY);
            }
            const bool maybe = mx * o > conf_pre;
            if ((o > conf) & has_cls & maybe) {                     // general.py:656,675 and a bound on general.py:691,698
                const int slot = atomicAdd(&q_n[w], 1);
                if (slot < STREAM_QUEUE) {
                    qx[w][slot] = x[u];
                    qm[w][slot] = make_uint2(__float_as_uint(o), vi[u]);
                } else {                                            // queue full: a tile of nothing but candidates on top
                    stream_direct<RAW, HALF>(x[u], o, vi[u] & ((1u << STREAM_SEG_SHIFT) - 1), c0[u], P.no, conf, class_keep,
                                       s_row_first[vi[u] >> STREAM_SEG_SHIFT], nc, &counts[b], cap, out);
                }                                                   // of what was waiting
            }
        }
        // the wave's own queue: LDS operations of one wave complete in order, so every lane reads the same count
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int n = __builtin_amdgcn_readfirstlane(q_n[w]);
        if (n >= 64) {                                              // enough to keep every lane busy
            drain(min(n, STREAM_QUEUE));
            if (lane == 0) q_n[w] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };
    // the workgroup's tiles, the next one's loads in flight while this one is looked at (two register sets, swapped by hand).
    // The loads of a tile are issued whether it exists or not: a branch around them would make the loop wait for the
    // newest loads in flight, not the oldest, where the two paths meet.
    dh_f4 xa[STREAM_U], xb[STREAM_U];
    float oa[STREAM_U], ob[STREAM_U];
    uint32_t ca[STREAM_U], cb[STREAM_U], va[STREAM_U], vb[STREAM_U];
    issue(t0, xa, oa, ca, va);
#pragma unroll 1
    for (uint32_t t = t0; t < t1; t += 2 * step) {
        issue(t + step, xb, ob, cb, vb);
        sift(xa, oa, ca, va);
        issue(t + 2 * step, xa, oa, ca, va);
        sift(xb, ob, cb, vb);
    }
    {
        const int n = __builtin_amdgcn_readfirstlane(q_n[w]);
        if (n > 0) drain(min(n, STREAM_QUEUE));
    }
    __syncthreads();
    const int n = min(s_n, s_valid);                                // the staged keys
    if (n == 0) return;                                             // workgroup-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(&counts[b], n);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if ((uint32_t)(s_base + i) < (uint32_t)cap) out[s_base + i] = stage[i];
}

// ------------------------------------------------------------------------------------------ 2/4 sort
__device__ __forceinline__ void cmpswap(u64& a, u64& b, bool ascending)
{
    if ((a > b) == ascending) { const u64 t = a; a = b; b = t; }
}

// R consecutive steps (j = 2^h, 2^(h-1), ..., 2^(h-R+1)) of stage k of the bitonic network in one
// LDS pass: a thread pulls the 2^R keys that differ only in those R index bits into registers,
// runs the R compare-exchange steps there, and writes them back.  All 2^R keys share the stage
// direction because bit k lies above every j bit.  This cuts LDS traffic (the limiter of an
// LDS-resident bitonic sort) by R.
template <int R>
__device__ __forceinline__ void bitonic_pass(u64* s, uint32_t chunk, uint32_t gbase, uint32_t k, uint32_t h)
{
    const uint32_t low = h + 1 - R;                       // lowest of the R bit positions
    const uint32_t lowmask = (1u << low) - 1u;
    for (uint32_t t = threadIdx.x; t < (chunk >> R); t += blockDim.x) {
        const uint32_t base = ((t & ~lowmask) << R) | (t & lowmask);
        u64 v[1 << R];
#pragma unroll
        for (int q = 0; q < (1 << R); ++q) v[q] = s[SIDX(base | ((uint32_t)q << low))];
        const bool asc = ((gbase + base) & k) == 0;
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
#pragma unroll
            for (int q = 0; q < (1 << R); ++q) {
                if (!(q & (1 << r))) {
                    const u64 a = v[q], c = v[q | (1 << r)];
                    const bool sw = (a > c) == asc;
                    v[q] = sw ? c : a;
                    v[q | (1 << r)] = sw ? a : c;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < (1 << R); ++q) s[SIDX(base | ((uint32_t)q << low))] = v[q];
    }
    __syncthreads();
}

// all (k, j) steps of the bitonic network with j < chunk, on `chunk` keys held in LDS (key i at s[SIDX(i)]);
// `gbase` is the global position of s[0] (it fixes the direction bits for k > chunk).
__device__ void bitonic_lds(u64* s, uint32_t chunk, uint32_t gbase, uint32_t k_first, uint32_t k_last)
{
    for (uint32_t k = k_first; k <= k_last; k <<= 1) {
        const uint32_t jtop = (k >> 1) < chunk ? (k >> 1) : (chunk >> 1);
        int h = 31 - __clz(jtop);                         // bit position of the first step's j
        while (h >= 0) {
            if (h >= 3 && chunk >= 16) { bitonic_pass<4>(s, chunk, gbase, k, (uint32_t)h); h -= 4; }
            else if (h == 2 && chunk >= 8) { bitonic_pass<3>(s, chunk, gbase, k, 2u); h -= 3; }
            else if (h == 1 && chunk >= 4) { bitonic_pass<2>(s, chunk, gbase, k, 1u); h -= 2; }
            else { bitonic_pass<1>(s, chunk, gbase, k, (uint32_t)h); h -= 1; }
        }
    }
}

// Bitonic sort of npad <= 1024 keys (a power of two, key i at s[SIDX(i)]) by a 1024-thread workgroup, one key per thread
// in a register: partners less than 64 apart exchange by lane shuffle (no barrier, every wave busy), the others through
// LDS.  For these sizes the passes of bitonic_lds keep a single wave busy between barriers.
__device__ void bitonic_small(u64* s, uint32_t npad)
{
    const uint32_t t = threadIdx.x;
    u64 key = t < npad ? s[SIDX(t)] : KEY_SENTINEL;
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        const bool asc = (t & k) == 0;
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            u64 other;
            if (j >= 64) {
                __syncthreads();
                if (t < npad) s[SIDX(t)] = key;
                __syncthreads();
                other = t < npad ? s[SIDX(t ^ j)] : KEY_SENTINEL;
            } else {
                const uint32_t lo = __shfl_xor((uint32_t)key, (int)j, 64), hi = __shfl_xor((uint32_t)(key >> 32), (int)j, 64);
                other = ((u64)hi << 32) | lo;
            }
            const bool low_side = (t & j) == 0;                 // this thread keeps the smaller key of the pair when ascending
            const bool take_min = low_side == asc;
            key = take_min ? (other < key ? other : key) : (other > key ? other : key);
        }
    }
    __syncthreads();
    if (t < npad) s[SIDX(t)] = key;
    __syncthreads();
}

__device__ __forceinline__ uint32_t pow2_ceil(uint32_t n)
{
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// first/last position of every class in a sorted [class:32 | rank:32] key array -> seg[class] = {begin, end}
template <bool PADDED>
__device__ void write_segments(const u64* k, int n, int2* __restrict__ seg)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t c = (uint32_t)(k[PADDED ? SIDX(i) : i] >> 32);
        if (i == 0) seg[c].x = 0;
        else {
            const uint32_t prev = (uint32_t)(k[PADDED ? SIDX(i - 1) : i - 1] >> 32);
            if (prev != c) { seg[c].x = i; seg[prev].y = i; }
        }
        if (i == n - 1) seg[c].y = n;
    }
}

// One workgroup sorts one image's keys ascending.  n = min(counts[b], limit).
// seg_all != nullptr: the keys are [class | rank]; also record each class's [begin, end) (zero-initialised by the caller).
// m_out != nullptr: also record how many of the sorted keys go on to NMS, min(n, max_nms) (general.py:716-717).
__global__ void __launch_bounds__(SORT_THREADS) k_sort_keys(u64* __restrict__ keys_all, const int* __restrict__ counts,
                                                            int limit, int cap_pad, int2* __restrict__ seg_all, int nc,
                                                            int* __restrict__ m_out, int max_nms)
{
    __shared__ u64 s[SORT_LDS_SLOTS(SORT_LDS_KEYS)];
    u64* __restrict__ keys = keys_all + (size_t)blockIdx.x * cap_pad;
    int2* __restrict__ seg = seg_all ? seg_all + (size_t)blockIdx.x * nc : nullptr;
    int n = counts[blockIdx.x];
    if (n > limit) n = limit;
    if (m_out && threadIdx.x == 0) m_out[blockIdx.x] = n < max_nms ? (n > 0 ? n : 0) : max_nms;
    if (n <= 0) return;
    if (n == 1) {
        if (seg && threadIdx.x == 0) { const uint32_t c = (uint32_t)(keys[0] >> 32); seg[c].x = 0; seg[c].y = 1; }
        return;
    }
    const uint32_t npad = pow2_ceil((uint32_t)n);
    if (npad <= SORT_LDS_KEYS) {
        for (uint32_t i = threadIdx.x; i < npad; i += blockDim.x) s[SIDX(i)] = (i < (uint32_t)n) ? keys[i] : KEY_SENTINEL;
        __syncthreads();
        bitonic_lds(s, npad, 0, 2, npad);
        for (uint32_t i = threadIdx.x; i < (uint32_t)n; i += blockDim.x) keys[i] = s[SIDX(i)];
        if (seg) write_segments<true>(s, n, seg);
        return;
    }
    // larger than LDS: pad in global memory, sort LDS-sized chunks, then merge with global steps + LDS tails
    for (uint32_t i = n + threadIdx.x; i < npad; i += blockDim.x) keys[i] = KEY_SENTINEL;
    __syncthreads();
    for (uint32_t base = 0; base < npad; base += SORT_LDS_KEYS) {
        for (uint32_t i = threadIdx.x; i < SORT_LDS_KEYS; i += blockDim.x) s[SIDX(i)] = keys[base + i];
        __syncthreads();
        bitonic_lds(s, SORT_LDS_KEYS, base, 2, SORT_LDS_KEYS);
        for (uint32_t i = threadIdx.x; i < SORT_LDS_KEYS; i += blockDim.x) keys[base + i] = s[SIDX(i)];
        __syncthreads();
    }
    for (uint32_t k = 2 * SORT_LDS_KEYS; k <= npad; k <<= 1) {
        for (uint32_t j = k >> 1; j >= SORT_LDS_KEYS; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < npad / 2; t += blockDim.x) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i | j;
                const bool asc = (i & k) == 0;
                const u64 a = keys[i], c = keys[l];
                if ((a > c) == asc) { keys[i] = c; keys[l] = a; }
            }
            __syncthreads();   // one workgroup owns the image: its own global writes are visible to it after the barrier
        }
        for (uint32_t base = 0; base < npad; base += SORT_LDS_KEYS) {
            for (uint32_t i = threadIdx.x; i < SORT_LDS_KEYS; i += blockDim.x) s[SIDX(i)] = keys[base + i];
            __syncthreads();
            bitonic_lds(s, SORT_LDS_KEYS, base, k, k);
            for (uint32_t i = threadIdx.x; i < SORT_LDS_KEYS; i += blockDim.x) keys[base + i] = s[SIDX(i)];
            __syncthreads();
        }
    }
    if (seg) write_segments<false>(keys, n, seg);
}

// ------------------------------------------------------------------------------------------ 2' select + sort
// Only the rows that can reach the output need to go through NMS: the output is the first max_det kept rows in score
// order, and whether a row is kept depends only on rows that rank above it.  So NMS on a score-ordered prefix is
// exact for that prefix, and if the prefix yields max_det kept rows nothing below it matters.  This kernel picks
// the prefix: the k_target best keys of the image plus everything that shares the 22-bit score prefix of the last
// one (two rounds of an 11-bit radix select in LDS), sorted.  k_nms_collect reports the images whose prefix did not
// yield max_det rows (needs_full); the caller repeats those batches with the full sort.
#define SELECT_BINS 2048

__global__ void __launch_bounds__(SORT_THREADS) k_select_sort(const u64* __restrict__ keys_all, const int* __restrict__ counts,
                                                              int limit, int cap_pad, int k_target, int max_nms,
                                                              u64* __restrict__ top_all, int top_pad,
                                                              int* __restrict__ m_out, int* __restrict__ truncated)
{
    __shared__ u64 s[SORT_LDS_SLOTS(SORT_LDS_KEYS)];
    __shared__ int hist[SELECT_BINS];
    __shared__ int wsum[SORT_THREADS / 64];
    __shared__ int s_bin, s_below, s_count, s_fill;
    const u64* __restrict__ keys = keys_all + (size_t)blockIdx.x * cap_pad;
    u64* __restrict__ top = top_all + (size_t)blockIdx.x * top_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n = counts[blockIdx.x];
    if (n > limit) n = limit;
    if (n <= 0) {
        if (threadIdx.x == 0) { m_out[blockIdx.x] = 0; truncated[blockIdx.x] = 0; }
        return;
    }
    const int n_lim = n < max_nms ? n : max_nms;
    int T = n;
    if (n > k_target) {
        uint32_t prefix = 0;
        int below = 0;
        for (int round = 0; round < 2; ++round) {
            const int shift = round == 0 ? 53 : 42;
            for (int i = threadIdx.x; i < SELECT_BINS; i += SORT_THREADS) hist[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += SORT_THREADS) {
                const u64 key = keys[i];
                if (round == 0 || (uint32_t)(key >> 53) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & (SELECT_BINS - 1)], 1);
            }
            __syncthreads();
            // the bin that holds the (k_target - below)-th key of this round: block-wide inclusive scan, 2 bins per thread
            const int c0 = hist[2 * threadIdx.x], c1 = hist[2 * threadIdx.x + 1];
            int incl = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            incl += base;
            const int excl = incl - (c0 + c1);
            const int need = k_target - below;
            if (excl < need && need <= excl + c0) { s_bin = 2 * threadIdx.x; s_below = below + excl; s_count = c0; }
            else if (excl + c0 < need && need <= incl) { s_bin = 2 * threadIdx.x + 1; s_below = below + excl + c0; s_count = c1; }
            __syncthreads();
            prefix = round == 0 ? (uint32_t)s_bin : ((prefix << 11) | (uint32_t)s_bin);
            below = s_below;
            T = s_below + s_count;
            __syncthreads();
        }
        if (T > SORT_LDS_KEYS) {      // too many keys share one 22-bit score prefix: leave this image to the full path
            if (threadIdx.x == 0) { m_out[blockIdx.x] = 0; truncated[blockIdx.x] = 1; }
            return;
        }
        if (threadIdx.x == 0) s_fill = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += SORT_THREADS) {
            const int i = i0 + threadIdx.x;
            u64 key = 0;
            bool take = false;
            if (i < n) { key = keys[i]; take = (uint32_t)(key >> 42) <= prefix; }
            const u64 bm = __ballot(take);
            int slot = 0;
            if (lane == 0 && bm) slot = atomicAdd(&s_fill, __popcll(bm));
            slot = __shfl(slot, 0, 64) + __popcll(bm & lanemask_lt());
            if (take) s[SIDX(slot)] = key;
        }
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < n; i += SORT_THREADS) s[SIDX(i)] = keys[i];
    }
    const uint32_t npad = pow2_ceil((uint32_t)(T < 2 ? 2 : T));
    for (uint32_t i = T + threadIdx.x; i < npad; i += SORT_THREADS) s[SIDX(i)] = KEY_SENTINEL;
    __syncthreads();
    bitonic_lds(s, npad, 0, 2, npad);
    for (int i = threadIdx.x; i < T; i += SORT_THREADS) top[i] = s[SIDX(i)];
    if (threadIdx.x == 0) {
        m_out[blockIdx.x] = T < max_nms ? T : max_nms;
        truncated[blockIdx.x] = T < n_lim ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ 3 gather
// monotone float <-> uint32 (ascending); both range words below are kept with atomicMax so that an
// all-zero workspace is their valid initial state: hi_key = max key(hi), lo_inv = max ~key(lo)
__device__ __forceinline__ uint32_t float_ukey(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ukey_float(uint32_t e)
{
    return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e);
}

struct img_range {   // per image: coordinate range of the boxes that reach NMS (+ a not-finite flag)
    uint32_t hi_key, lo_inv;
    int bad, pad;
};

template <bool RAW, typename T = float>
__global__ void __launch_bounds__(256) k_nms_gather(const nms_source S, int A, dh_fastdiv ncd,
                                                    const u64* __restrict__ keys, int cap_pad,
                                                    const int* __restrict__ m_arr,
                                                    float* __restrict__ det, u64* __restrict__ keys2, int m_pad,
                                                    int m_cap, img_range* __restrict__ range)
{
    const int b = blockIdx.y;
    const int m = m_arr[b];                                                         // rows that go on to NMS
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int no = (int)ncd.d + 5;
    float lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    if (r < m) {
        const u64 key = keys[(size_t)b * cap_pad + r];
        const uint32_t ord = (uint32_t)key;
        uint32_t anchor = dh_div(ord, ncd);
        const uint32_t cls = ord - anchor * ncd.d;
        if (anchor >= (uint32_t)A) anchor = (uint32_t)A - 1;      // a key always names a row of the image: never address beyond it
        float x1, y1, x2, y2;                                                           // general.py:553-560
        if (RAW) {
            const float4 c = raw_corners<T>(S, b, anchor);
            x1 = c.x; y1 = c.y; x2 = c.z; y2 = c.w;
        } else {
            const float4 c = pred_corners<T>(S, b, A, anchor, no);
            x1 = c.x; y1 = c.y; x2 = c.z; y2 = c.w;
        }
        float* __restrict__ d = det + ((size_t)b * m_cap + r) * 6;
        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
        d[4] = key_score((uint32_t)(key >> 32));
        d[5] = (float)cls;
        keys2[(size_t)b * m_pad + r] = ((u64)cls << 32) | (uint32_t)r;
        lo = fminf(fminf(x1, y1), fminf(x2, y2));
        hi = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
        bad = !(fabsf(x1) < 1e30f && fabsf(y1) < 1e30f && fabsf(x2) < 1e30f && fabsf(y2) < 1e30f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    __shared__ float s_lo[4], s_hi[4];
    __shared__ int s_bad[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_bad[wave] = __ballot(bad) != 0; }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x * blockDim.x < m) {
        lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
        hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
        // The current range is read from L2 (agent scope; a CU's L1 would keep returning the initial
        // value) and the atomic is issued only if this block widens it, which is rare after the first few.
        if (lo <= hi) {
            const uint32_t lk = ~float_ukey(lo), hk = float_ukey(hi);
            if (lk > __hip_atomic_load(&range[b].lo_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&range[b].lo_inv, lk);
            if (hk > __hip_atomic_load(&range[b].hi_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&range[b].hi_key, hk);
        }
        if (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) atomicOr(&range[b].bad, 1);
    }
}

// ------------------------------------------------------------------------------------------ 5 greedy
struct box_t {
    float x1, y1, x2, y2;
};

// suppress iff inter/(area_a + area_b - inter) > thr, evaluated exactly as the fp32 reference does;
// `thr` is the largest float <= the double threshold, which makes the fp32 ">" equal to the
// reference's comparison in double.  inter == 0 can never exceed a threshold >= 0, so only overlapping pairs count -- and
// of those the IEEE division (ten instructions) is evaluated only for the ones a reciprocal cannot decide: inter *
// rcp(union) is within 3e-7 (relative) of the quotient, so a value more than 1e-6 above or below thr settles the
// comparison of the correctly rounded quotient as well.  What lies in between, and pairs whose operands leave the
// range where that bound holds, take the division.
struct iou_thr {
    float thr, hi, lo;      // hi / lo: the reciprocal's verdict is final above / below these (+-inf: never final)
};

__device__ __forceinline__ iou_thr make_iou_thr(float thr)
{
    iou_thr t;
    t.thr = thr;
    const bool usable = thr > 1e-30f;
    t.hi = usable ? thr * 1.000001f : INFINITY;
    t.lo = usable ? thr * 0.999999f : -INFINITY;
    return t;
}

// Lanes (with `consider` set) whose box b is suppressed by box a.  Box a is the same in every lane (one kept box against
// a wave of candidates): straight-line code, the division in a branch that the whole wave skips unless some lane needs it.
__device__ __forceinline__ u64 suppressed_lanes(const box_t a, float area_a, const box_t b, float area_b, const iou_thr t, bool consider)
{
    const float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f);
    const float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
    const float inter = w * h;
    const float uni = area_a + area_b - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    const bool pos = consider & (inter > 0.0f);
    const bool normal = (uni > 1e-30f) & (uni < 1e30f) & (inter < 1e30f);
    const bool yes = pos & normal & (q > t.hi);
    const bool no = !pos | (normal & (q < t.lo));
    u64 m = __ballot(yes);
    if (__ballot(!yes & !no) != 0) m |= __ballot(!yes & !no & (__fdiv_rn(inter, uni) > t.thr));      // wave-uniform, rare
    return m;
}

// value of `v` in lane `src` (wave-uniform src): a VALU read into a scalar register, no LDS round trip
__device__ __forceinline__ float lane_value(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// WAVES waves per workgroup.  grid = (number of classes or 1, images).
//   class mode: block (c, b) owns the positions of class c in keys2 (sorted by class, then rank);
//   whole mode (agnostic, or an image whose coordinate range does not rule out cross-class overlap):
//   block (0, b) owns ranks 0..m-1 in rank order, each box shifted by its own class offset.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_nms_greedy(const float* __restrict__ det, const u64* __restrict__ keys2,
                                                           int m_pad, int m_cap, const int* __restrict__ m_arr,
                                                           const img_range* __restrict__ range,
                                                           const int2* __restrict__ seg_all, int agnostic, float max_wh,
                                                           float thr, int max_det,
                                                           unsigned char* __restrict__ keep, float4* __restrict__ kept_spill)
{
    const int b = blockIdx.y;
    const int m = m_arr[b];
    if (m == 0) return;
    bool whole = agnostic != 0;
    if (!whole && max_wh >= 0.0f) {      // max_wh < 0: every class is its own unshifted problem (plain per-class nms)
        const img_range rg = range[b];
        const float lo = ukey_float(~rg.lo_inv), hi = ukey_float(rg.hi_key);
        whole = rg.bad != 0 || !(hi - lo < max_wh - 1.0f);
    }
    int s, e;
    const u64* __restrict__ k2 = keys2 + (size_t)b * m_pad;
    if (whole) {
        if (blockIdx.x != 0) return;
        s = 0; e = m;
    } else {
        const int2 sg = seg_all[(size_t)b * gridDim.x + blockIdx.x];
        s = sg.x < 0 ? 0 : sg.x;
        e = sg.y > m ? m : sg.y;          // a segment lies inside the image's m rows whatever the table holds
        if (s >= e) return;
    }
    const float* __restrict__ dimg = det + (size_t)b * m_cap * 6;
    unsigned char* __restrict__ kimg = keep + (size_t)b * m_cap;
    float4* __restrict__ spill = kept_spill + (size_t)b * m_cap + s;   // kept boxes of this segment beyond the LDS part

    __shared__ float4 kbox[KEPT_LDS];
    __shared__ float karea[KEPT_LDS];
    __shared__ u64 sup_mask[WAVES];
    __shared__ float4 cbox[64];      // the 64 candidates of the current chunk
    __shared__ float carea[64];
    __shared__ u64 cmask[64];
    __shared__ int s_kept;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float off_scale = (agnostic || max_wh < 0.0f) ? 0.0f : max_wh;
    const iou_thr TH = make_iou_thr(thr);
    int kept = 0;

    for (int base = s; base < e && kept < max_det; base += 64) {
        const int p = base + lane;
        const bool valid = p < e;
        int rank = 0;
        box_t bx = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            rank = whole ? p : (int)(uint32_t)k2[p];
            const float* __restrict__ d = dimg + (size_t)rank * 6;
            const float c = d[5] * off_scale;                                     // general.py:720
            bx.x1 = d[0] + c; bx.y1 = d[1] + c; bx.x2 = d[2] + c; bx.y2 = d[3] + c;   // general.py:721
        }
        const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
        // phase 1: against everything kept so far, kept list split across the waves
        u64 supm = 0;
        for (int k = wave; k < kept; k += WAVES) {
            float4 kb;
            float ka;
            if (k < KEPT_LDS) { kb = kbox[k]; ka = karea[k]; }
            else { kb = spill[k]; ka = (kb.z - kb.x) * (kb.w - kb.y); }
            const box_t kbb = {kb.x, kb.y, kb.z, kb.w};
            supm |= suppressed_lanes(kbb, ka, bx, area, TH, true);
        }
        const u64 sm = supm;
        if (lane == 0) sup_mask[wave] = sm;
        if (wave == 0) { cbox[lane] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2); carea[lane] = area; }
        __syncthreads();
        // phase 2a: all waves build the chunk's 64x64 suppression bit matrix, 64/WAVES rows each
        // (row i = candidates j > i that box i would suppress)
#pragma unroll 4
        for (int q = 0; q < 64 / WAVES; ++q) {
            const int i = wave * (64 / WAVES) + q;
            const float4 bi4 = cbox[i];
            const box_t bi = {bi4.x, bi4.y, bi4.z, bi4.w};
            const u64 hm = suppressed_lanes(bi, carea[i], bx, area, TH, lane > i);
            if (lane == 0) cmask[i] = hm;
        }
        __syncthreads();
        // phase 2b: wave 0 walks the chunk in score order with scalar mask arithmetic
        if (wave == 0) {
            u64 dead = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) dead |= sup_mask[w];
            u64 todo = __ballot(valid) & ~dead;
            u64 kept_mask = 0;
            int room = max_det - kept;
            while (todo && room > 0) {
                const int i = __ffsll((long long)todo) - 1;
                kept_mask |= 1ull << i;
                todo &= ~(cmask[i] | (1ull << i));
                --room;
            }
            if ((kept_mask >> lane) & 1ull) {
                const int slot = kept + __popcll(kept_mask & lanemask_lt());
                kimg[rank] = 1;
                const float4 v = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
                if (slot < KEPT_LDS) { kbox[slot] = v; karea[slot] = area; }
                else spill[slot] = v;
            }
            if (lane == 0) s_kept = kept + __popcll(kept_mask);
        }
        __syncthreads();
        kept = s_kept;
    }
}

// ------------------------------------------------------------------------------------------ 6 collect
// First max_det kept ranks (ascending rank == descending score) -> out rows (or original indices).
// Rows of all images are packed back to back: a workgroup first counts its image's rows, reserves
// [base, base+count) with one atomicAdd on *cursor, then writes; out_span[b] = {base, count}.
// (cursor == nullptr: single image, base 0.)
__global__ void __launch_bounds__(1024) k_nms_collect(const float* __restrict__ det, const unsigned char* __restrict__ keep,
                                                      const u64* __restrict__ keys, int cap_pad, int m_cap,
                                                      const int* __restrict__ counts, const int* __restrict__ m_arr,
                                                      const int* __restrict__ truncated, int max_det,
                                                      float* __restrict__ out, long long* __restrict__ out_index,
                                                      int* __restrict__ out_count, int* __restrict__ cursor,
                                                      int2* __restrict__ out_span, int* __restrict__ cand_count,
                                                      int* __restrict__ needs_full, int out_rows)
{
    const int b = blockIdx.x;
    if (cand_count && threadIdx.x == 0) cand_count[b] = counts[b];
    const int m = m_arr[b];
    __shared__ int wcount[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* __restrict__ kimg = keep + (size_t)b * m_cap;
    // pass 1: how many rows this image contributes
    int total = 0;
    for (int base = 0; base < m && total < max_det; base += 1024) {
        const int r = base + threadIdx.x;
        const u64 bm = __ballot(r < m && kimg[r] != 0);
        if (lane == 0) wcount[wave] = __popcll(bm);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; ++w) total += wcount[w];
        __syncthreads();
    }
    // Only a score-ordered prefix went through NMS (truncated): its kept rows are the final answer iff there are
    // max_det of them, because rows beyond the prefix rank below every one of these.
    if (needs_full && threadIdx.x == 0) needs_full[b] = (truncated && truncated[b] && total < max_det) ? 1 : 0;
    if (total > max_det) total = max_det;
    if (threadIdx.x == 0) {
        const int first = cursor ? atomicAdd(cursor, total) : 0;
        s_base = first;
        if (out_count) out_count[b] = total;
        if (out_span) out_span[b] = make_int2(first, total);
    }
    __syncthreads();
    const int first = s_base;
    // pass 2: write them in rank order
    int done = 0;
    for (int base = 0; base < m && done < total; base += 1024) {
        const int r = base + threadIdx.x;
        const bool k = r < m && kimg[r] != 0;
        const u64 bm = __ballot(k);
        if (lane == 0) wcount[wave] = __popcll(bm);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = wcount[w]; if (w < wave) before += c; all += c; }
        const int pos = done + before + __popcll(bm & lanemask_lt());
        if (k && pos < total && (long long)first + pos < out_rows) {      // never beyond the caller's buffer, whatever the cursor holds
            if (out) {
                const float* __restrict__ d = det + ((size_t)b * m_cap + r) * 6;
                float* __restrict__ o = out + ((size_t)first + pos) * 6;
#pragma unroll
                for (int q = 0; q < 6; ++q) o[q] = d[q];
            }
            if (out_index) out_index[(size_t)first + pos] = (long long)(uint32_t)keys[(size_t)b * cap_pad + r];
        }
        done += all;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ 2-6 in one workgroup
// The usual inference case has a few dozen candidates per image: six launches of ~10 us each are then mostly launch
// latency.  This kernel does everything after the prefilter for an image with at most SMALL_MAX candidates in one
// workgroup and LDS only: sort the keys, fetch the boxes, greedy NMS in score order on boxes shifted by their class
// offset (exactly what the reference hands to torchvision: general.py:720-722), first max_det kept rows out.
// An image with more candidates sets needs_full and is left alone: the caller repeats the batch on the general path.
#define SMALL_MAX 1024

template <bool RAW, typename T = float>
__global__ void __launch_bounds__(1024) k_nms_small(const nms_source S, int A, dh_fastdiv ncd, const u64* __restrict__ keys_all,
                                                    int cap_pad, const int* __restrict__ counts, int cap, int agnostic,
                                                    float max_wh, float thr, int max_det, float* __restrict__ out,
                                                    int* __restrict__ cursor, int2* __restrict__ out_span,
                                                    int* __restrict__ cand_count, int* __restrict__ needs_full, int out_rows)
{
    __shared__ u64 s_key[SORT_LDS_SLOTS(SMALL_MAX)];      // key i at SIDX(i)
    __shared__ float4 s_box[SMALL_MAX];         // unshifted x1,y1,x2,y2 in score order
    __shared__ float s_off[SMALL_MAX];          // class * max_wh (0 when agnostic)
    __shared__ float4 kbox[SMALL_MAX];          // kept boxes (shifted)
    __shared__ float karea[SMALL_MAX];
    __shared__ unsigned char s_keep[SMALL_MAX];
    __shared__ u64 sup_mask[16], cmask[64];
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    __shared__ int s_kept, s_base, wcount[16];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_all = counts[b];
    if (threadIdx.x == 0) cand_count[b] = n_all;
    if (n_all > SMALL_MAX || n_all > cap) {                         // workgroup-uniform
        if (threadIdx.x == 0) { needs_full[b] = 1; out_span[b] = make_int2(0, 0); }
        return;
    }
    const int n = n_all;
    // sort by score (descending scores are ascending keys)
    const uint32_t npad = pow2_ceil((uint32_t)(n < 2 ? 2 : n));
    for (uint32_t i = threadIdx.x; i < npad; i += 1024) s_key[SIDX(i)] = (int)i < n ? keys_all[(size_t)b * cap_pad + i] : KEY_SENTINEL;
    s_keep[threadIdx.x] = 0;
    __syncthreads();
    bitonic_lds(s_key, npad, 0, 2, npad);
    // fetch the boxes (general.py:553-560 xywh -> xyxy on the decoded row)
    const int no = (int)ncd.d + 5;
    if ((int)threadIdx.x < n) {
        const uint32_t ord = (uint32_t)s_key[SIDX(threadIdx.x)];
        uint32_t anchor = dh_div(ord, ncd);
        const uint32_t cls = ord - anchor * ncd.d;
        if (anchor >= (uint32_t)A) anchor = (uint32_t)A - 1;      // a key always names a row of the image: never address beyond it
        if (RAW) {
            s_box[threadIdx.x] = raw_corners<T>(S, b, anchor);
        } else {
            s_box[threadIdx.x] = pred_corners<T>(S, b, A, anchor, no);
        }
        s_off[threadIdx.x] = agnostic ? 0.0f : (float)cls * max_wh;                      // general.py:720
    }
    __syncthreads();
    // greedy NMS in score order, 64 candidates at a time (the scheme of k_nms_greedy, whole-image mode)
    const iou_thr TH = make_iou_thr(thr);
    int kept = 0;
    for (int base = 0; base < n && kept < max_det; base += 64) {
        const int p = base + lane;
        const bool valid = p < n;
        box_t bx = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const float4 v = s_box[p];
            const float c = s_off[p];
            bx.x1 = v.x + c; bx.y1 = v.y + c; bx.x2 = v.z + c; bx.y2 = v.w + c;          // general.py:721
        }
        const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
        u64 supm = 0;
        for (int k = wave; k < kept; k += 16) {
            const float4 kb = kbox[k];
            const box_t kbb = {kb.x, kb.y, kb.z, kb.w};
            supm |= suppressed_lanes(kbb, karea[k], bx, area, TH, true);
        }
        const u64 sm = supm;
        if (lane == 0) sup_mask[wave] = sm;
        if (wave == 0) { cbox[lane] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2); carea[lane] = area; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = wave * 4 + q;
            const float4 bi4 = cbox[i];
            const box_t bi = {bi4.x, bi4.y, bi4.z, bi4.w};
            const u64 hm = suppressed_lanes(bi, carea[i], bx, area, TH, lane > i);
            if (lane == 0) cmask[i] = hm;
        }
        __syncthreads();
        if (wave == 0) {
            u64 dead = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) dead |= sup_mask[w];
            u64 todo = __ballot(valid) & ~dead;
            u64 kept_mask = 0;
            int room = max_det - kept;
            while (todo && room > 0) {
                const int i = __ffsll((long long)todo) - 1;
                kept_mask |= 1ull << i;
                todo &= ~(cmask[i] | (1ull << i));
                --room;
            }
            if ((kept_mask >> lane) & 1ull) {
                const int slot = kept + __popcll(kept_mask & lanemask_lt());
                s_keep[p] = 1;
                kbox[slot] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
                karea[slot] = area;
            }
            if (lane == 0) s_kept = kept + __popcll(kept_mask);
        }
        __syncthreads();
        kept = s_kept;
    }
    // the kept rows, in score order, into this image's span of the packed output
    const int total = kept < max_det ? kept : max_det;
    if (threadIdx.x == 0) {
        const int first = atomicAdd(cursor, total);
        s_base = first;
        out_span[b] = make_int2(first, total);
        needs_full[b] = 0;
    }
    const bool k = (int)threadIdx.x < n && s_keep[threadIdx.x] != 0;
    const u64 bm = __ballot(k);
    if (lane == 0) wcount[wave] = __popcll(bm);
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) if (w < wave) before += wcount[w];
    const int pos = before + __popcll(bm & lanemask_lt());
    if (k && pos < total && (long long)s_base + pos < out_rows) {
        const u64 key = s_key[SIDX(threadIdx.x)];
        const uint32_t ord = (uint32_t)key;
        const uint32_t anchor = dh_div(ord, ncd);
        const float4 v = s_box[threadIdx.x];
        float* __restrict__ o = out + ((size_t)s_base + pos) * 6;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        o[4] = key_score((uint32_t)(key >> 32));
        o[5] = (float)(ord - anchor * ncd.d);
    }
}

// ------------------------------------------------------------------------------------------ 2'-6, class groups
// Validation thresholds leave thousands of candidates per image, of which only a score-ordered prefix can reach the
// output (see k_select_sort).  Select + sort, gather, class sort, greedy and collect as five launches spend most of their
// time at launch boundaries, on round trips through global memory for a few thousand rows, and in stages that one
// workgroup per image runs alone (64 of 256 CUs busy at bs 64).  Class-aware NMS falls apart by class, so this kernel
// gives an image G workgroups, each responsible for the classes c with c % G == g, and does everything in LDS:
//   a  every workgroup of the image histograms the scores of ALL its candidates (2048 bins, 128 per octave) and derives the
//      same cut: the best bins that together hold at least GRP_TARGET keys -- the selected prefix, T <= GRP_MAX keys;
//   b  it collects the selected keys of its own classes as [class : 9 | score order : 32 | anchor : 23] and sorts them:
//      every class becomes a contiguous list in score order (equal scores: ascending anchor, the reference's candidate order);
//   c  it fetches the boxes, runs greedy NMS per list (long lists by the whole workgroup one after another, short ones
//      by single waves side by side) and appends the kept rows to the image's list in global memory;
//   d  a second launch (k_nms_groups_final, one workgroup per image) picks the first max_det kept rows in score order and
//      writes them: the kernel boundary makes every workgroup's rows visible, which inside one launch would take an
//      agent-scope fence across the non-coherent L2s and cost more than the boundary does.
// needs_full[b] = 1, as after k_select_sort: the prefix did not yield max_det rows.  needs_full[b] = 2: the image is left
// to the separate launches (too many equal scores for the cut, more than max_nms keys selected, or a box outside the
// window in which the class offset alone rules out overlap across classes).
#define GRP_MAX 4096            // selected keys of an image (all its workgroups together); also what one workgroup's tables hold
#define GRP_TARGET 3584         // the cut is made where the best bins hold at least this many keys
#define GRP_BINS 2048
#define GRP_NC 512              // classes (9 bits of the sort key)
#define GRP_KEPT 512            // kept list of a list handled by the whole workgroup (>= max_det)
#define GRP_BIG 192             // class lists longer than this are handled by the whole workgroup
#define GRP_BIG_LISTS 24        // at most GRP_MAX / GRP_BIG of them (rounded up)
#define GRP_FINAL 1024          // rows sorted for the output (more candidates for it: the full GRP_MAX sort)
#define GRP_ANCHOR_BITS 23
#define GRP_BATCH 8             // key loads a thread keeps in flight while it walks an image's candidates

// Histogram bin of a key: 0 for scores >= 1, then 128 bins per octave going down, the last bin for everything below
// 2^-16 and for negative scores.  Monotone in the key (ascending key = descending score).
__device__ __forceinline__ uint32_t score_bin(uint32_t key_hi)
{
    const uint32_t e = ~key_hi;
    if (!(e & 0x80000000u)) return GRP_BINS - 1;
    const uint32_t sb = e & 0x7FFFFFFFu;
    if (sb >= 0x3F800000u) return 0;
    const uint32_t d = (0x3F800000u - sb) >> 16;
    return d < GRP_BINS - 1 ? d : GRP_BINS - 1;
}

// Smallest bin B whose cumulative count reaches `need` (or the last bin), and that cumulative count; hist has GRP_BINS
// entries, all 1024 threads call.  sh: 2 ints of LDS, wsum: 16.
__device__ void cut_bin(const int* hist, int need, int* wsum, int* sh, int& bin, int& upto)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = hist[2 * threadIdx.x], c1 = hist[2 * threadIdx.x + 1];
    int incl = c0 + c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    if (threadIdx.x == 0) { sh[0] = GRP_BINS - 1; sh[1] = -1; }
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    incl += base;
    const int excl = incl - (c0 + c1);
    if (excl < need && need <= excl + c0) { sh[0] = 2 * threadIdx.x; sh[1] = excl + c0; }
    else if (excl + c0 < need && need <= incl) { sh[0] = 2 * threadIdx.x + 1; sh[1] = incl; }
    __syncthreads();
    bin = sh[0];
    upto = sh[1];
    if (upto < 0) {                       // fewer than `need` keys in all: everything
        upto = 0;
        for (int w = 0; w < 16; ++w) upto += wsum[w];
    }
    __syncthreads();
}

template <bool RAW, typename T = float>
__global__ void __launch_bounds__(1024) k_nms_groups(const nms_source S, int A, dh_fastdiv ncd, const u64* __restrict__ keys_all,
                                                     int cap_pad, const int* __restrict__ counts, int cap, int max_nms, float max_wh,
                                                     float thr, int max_det, float* __restrict__ kept_all, int m_cap,
                                                     int* __restrict__ kept_count, int* __restrict__ truncated_out,
                                                     int* __restrict__ left_flag)
{
    __shared__ u64 s_key[SORT_LDS_SLOTS(GRP_MAX)];    // key i at SIDX(i)
    __shared__ float4 s_box[GRP_MAX];                 // unshifted x1,y1,x2,y2 of the sorted keys
    __shared__ unsigned char s_keep[GRP_MAX];
    __shared__ int hist[GRP_BINS];                    // score histogram; then [0,nc) list end, [nc,2nc) list start
    __shared__ float4 kbox[GRP_KEPT];
    __shared__ float karea[GRP_KEPT];
    __shared__ u64 sup_mask[16], cmask[64];
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    __shared__ int wsum[16], sh[4], s_big[GRP_BIG_LISTS], s_nbig, s_next, s_kept, s_base, s_fill, s_bad;
    const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = (int)ncd.d;
    const int n_all = counts[b];
    const int n = n_all > cap ? cap : n_all;
    const u64* __restrict__ keys = keys_all + (size_t)b * cap_pad;
    float* __restrict__ kept_rows = kept_all + (size_t)b * m_cap * 6;      // {score order, anchor*nc + class, x1, y1, x2, y2} per kept row
    const int n_lim = n < max_nms ? n : max_nms;                          // general.py:716-717
    int ns = 0;                  // selected keys of this workgroup's classes
    int Tn = n;                  // selected keys of the image
    bool left = false;           // leave the image to the separate launches
    if (n > 0) {
        // a: the cut
        int cut = GRP_BINS - 1;
        if (n > GRP_MAX) {
            for (int i = threadIdx.x; i < GRP_BINS; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * GRP_BATCH) {      // GRP_BATCH loads in flight per thread
                u64 kk[GRP_BATCH];
#pragma unroll
                for (int u = 0; u < GRP_BATCH; ++u) kk[u] = i0 + 1024 * u < n ? keys[i0 + 1024 * u] : 0;
#pragma unroll
                for (int u = 0; u < GRP_BATCH; ++u)
                    if (i0 + 1024 * u < n) atomicAdd(&hist[score_bin((uint32_t)(kk[u] >> 32))], 1);
            }
            __syncthreads();
            cut_bin(hist, GRP_TARGET, wsum, sh, cut, Tn);
            left = Tn > GRP_MAX || Tn > max_nms;
        } else {
            left = n > max_nms;
        }
        // b: this workgroup's share of the selected keys, sorted by (class, score order, anchor)
        if (threadIdx.x == 0) { s_fill = 0; s_next = 0; s_bad = 0; }
        for (int i = threadIdx.x; i < GRP_MAX; i += 1024) s_keep[i] = 0;
        for (int i = threadIdx.x; i < 2 * nc; i += 1024) hist[i] = 0;
        __syncthreads();
        if (!left) {
            for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * GRP_BATCH) {
                u64 kk[GRP_BATCH];
#pragma unroll
                for (int u = 0; u < GRP_BATCH; ++u) kk[u] = i0 + 1024 * u < n ? keys[i0 + 1024 * u] : 0;
#pragma unroll
                for (int u = 0; u < GRP_BATCH; ++u) {
                    const u64 key = kk[u];
                    const uint32_t ord = (uint32_t)key;
                    const uint32_t anchor = dh_div(ord, ncd);
                    const uint32_t cls = ord - anchor * ncd.d;
                    const bool take = i0 + 1024 * u < n && (int)score_bin((uint32_t)(key >> 32)) <= cut && (int)(cls % (uint32_t)G) == g;
                    const u64 ck = ((u64)cls << 55) | ((key >> 32) << GRP_ANCHOR_BITS) | (u64)(anchor & ((1u << GRP_ANCHOR_BITS) - 1u));
                    const u64 bm = __ballot(take);
                    if (bm == 0) continue;                              // wave-uniform
                    int slot = 0;
                    if (lane == 0) slot = atomicAdd(&s_fill, __popcll(bm));
                    slot = __builtin_amdgcn_readfirstlane(slot) + __popcll(bm & lanemask_lt());
                    if (take && slot < GRP_MAX) s_key[SIDX(slot)] = ck;
                }
            }
            __syncthreads();
            ns = s_fill < GRP_MAX ? s_fill : GRP_MAX;         // s_fill <= Tn <= GRP_MAX
        }
    }
    const bool truncated = Tn < n_lim;
    const iou_thr TH = make_iou_thr(thr);
    if (ns > 0) {
        const uint32_t npad = pow2_ceil((uint32_t)(ns < 2 ? 2 : ns));
        for (uint32_t i = ns + threadIdx.x; i < npad; i += 1024) s_key[SIDX(i)] = KEY_SENTINEL;
        __syncthreads();
        if (npad <= 1024) bitonic_small(s_key, npad); else bitonic_lds(s_key, npad, 0, 2, npad);
        // c: boxes (general.py:553-560 on the decoded row), list boundaries
        const int no = nc + 5;
        // every coordinate inside this window: boxes of different classes cannot overlap once shifted by class * max_wh
        const float win = 0.5f * (max_wh - 1.0f);
        bool bad = false;
        for (int i = threadIdx.x; i < ns; i += 1024) {
            const u64 ck = s_key[SIDX(i)];
            const int cls = (int)(ck >> 55);
            uint32_t anchor = (uint32_t)ck & ((1u << GRP_ANCHOR_BITS) - 1u);
            if (anchor >= (uint32_t)A) anchor = (uint32_t)A - 1;
            float x1, y1, x2, y2;
            if (RAW) {
                const float4 c = raw_corners<T>(S, b, anchor);
                x1 = c.x; y1 = c.y; x2 = c.z; y2 = c.w;
            } else {
                const float4 c = pred_corners<T>(S, b, A, anchor, no);
                x1 = c.x; y1 = c.y; x2 = c.z; y2 = c.w;
            }
            s_box[i] = make_float4(x1, y1, x2, y2);
            bad = bad || !(fabsf(x1) < win && fabsf(y1) < win && fabsf(x2) < win && fabsf(y2) < win);
            if (cls < nc) {
                if (i == 0 || (int)(s_key[SIDX(i - 1)] >> 55) != cls) hist[nc + cls] = i;
                if (i == ns - 1 || (int)(s_key[SIDX(i + 1)] >> 55) != cls) hist[cls] = i + 1;
            }
        }
        if (__ballot(bad) != 0 && lane == 0) s_bad = 1;
        __syncthreads();
        left = left || s_bad != 0;
        if (!left) {
            // lists the whole workgroup handles
            if (wave == 0) {
                int nbig_run = 0;                                       // wave-uniform
                for (int c0 = 0; c0 < nc; c0 += 64) {
                    const int c = c0 + lane;
                    const int len = c < nc ? hist[c] - hist[nc + c] : 0;
                    const u64 bigm = __ballot(len > GRP_BIG);
                    if (len > GRP_BIG) {
                        const int at = nbig_run + __popcll(bigm & lanemask_lt());
                        if (at < GRP_BIG_LISTS) s_big[at] = c;
                    }
                    nbig_run += __popcll(bigm);
                }
                if (lane == 0) s_nbig = nbig_run < GRP_BIG_LISTS ? nbig_run : GRP_BIG_LISTS;
            }
            __syncthreads();
            // greedy NMS, long lists: the whole workgroup, 64 candidates at a time (the scheme of k_nms_small)
            const int nbig = s_nbig;
            for (int q = 0; q < nbig; ++q) {
                const int c = s_big[q];
                const int start = hist[nc + c], len = hist[c] - start;
                const float off = (float)c * max_wh;                    // general.py:720
                int kept = 0;
                for (int base = 0; base < len && kept < max_det; base += 64) {
                    const int p = base + lane;
                    const bool valid = p < len;
                    box_t bx = {0.f, 0.f, 0.f, 0.f};
                    if (valid) {
                        const float4 v = s_box[start + p];
                        bx.x1 = v.x + off; bx.y1 = v.y + off; bx.x2 = v.z + off; bx.y2 = v.w + off;     // general.py:721
                    }
                    const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
                    u64 supm = 0;
                    for (int k = wave; k < kept; k += 16) {
                        const float4 kb = kbox[k];
                        const box_t kbb = {kb.x, kb.y, kb.z, kb.w};
                        supm |= suppressed_lanes(kbb, karea[k], bx, area, TH, true);
                    }
                    const u64 sm = supm;
                    if (lane == 0) sup_mask[wave] = sm;
                    if (wave == 0) { cbox[lane] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2); carea[lane] = area; }
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = wave * 4 + t;
                        const float4 bi4 = cbox[i];
                        const box_t bi = {bi4.x, bi4.y, bi4.z, bi4.w};
                        const u64 hm = suppressed_lanes(bi, carea[i], bx, area, TH, lane > i);
                        if (lane == 0) cmask[i] = hm;
                    }
                    __syncthreads();
                    if (wave == 0) {
                        u64 dead = 0;
#pragma unroll
                        for (int w = 0; w < 16; ++w) dead |= sup_mask[w];
                        u64 todo = __ballot(valid) & ~dead;
                        u64 kept_mask = 0;
                        int room = max_det - kept;
                        while (todo && room > 0) {
                            const int i = __ffsll((long long)todo) - 1;
                            kept_mask |= 1ull << i;
                            todo &= ~(cmask[i] | (1ull << i));
                            --room;
                        }
                        if ((kept_mask >> lane) & 1ull) {
                            const int slot = kept + __popcll(kept_mask & lanemask_lt());
                            s_keep[start + p] = 1;
                            kbox[slot] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
                            karea[slot] = area;
                        }
                        if (lane == 0) s_kept = kept + __popcll(kept_mask);
                    }
                    __syncthreads();
                    kept = s_kept;
                }
            }
            // short lists: one wave each, taken from a common counter; earlier kept boxes are found through the kept
            // masks of the list's earlier chunks (at most GRP_BIG / 64 of them)
            for (;;) {
                int ci = 0;
                if (lane == 0) ci = atomicAdd(&s_next, 1);
                const int c = g + __builtin_amdgcn_readfirstlane(ci) * G;
                if (c >= nc) break;
                const int start = hist[nc + c], len = hist[c] - start;
                if (len <= 0 || len > GRP_BIG) continue;
                const float off = (float)c * max_wh;                    // general.py:720
                u64 keptm[GRP_BIG / 64];
                int kept = 0;
#pragma unroll
                for (int j = 0; j < GRP_BIG / 64; ++j) {
                    keptm[j] = 0;
                    if (64 * j >= len || kept >= max_det) continue;     // wave-uniform
                    const int p = 64 * j + lane;
                    const bool valid = p < len;
                    box_t bx = {0.f, 0.f, 0.f, 0.f};
                    if (valid) {
                        const float4 v = s_box[start + p];
                        bx.x1 = v.x + off; bx.y1 = v.y + off; bx.x2 = v.z + off; bx.y2 = v.w + off;     // general.py:721
                    }
                    const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
                    u64 supm = 0;
#pragma unroll
                    for (int pj = 0; pj < j; ++pj) {
                        u64 km = keptm[pj];
                        while (km) {
                            const int i = __ffsll((long long)km) - 1;
                            km &= km - 1;
                            const float4 v = s_box[start + 64 * pj + i];
                            const box_t kb = {v.x + off, v.y + off, v.z + off, v.w + off};
                            supm |= suppressed_lanes(kb, (kb.x2 - kb.x1) * (kb.y2 - kb.y1), bx, area, TH, true);
                        }
                    }
                    u64 todo = __ballot(valid) & ~supm;
                    u64 kept_mask = 0;
                    int room = max_det - kept;
                    while (todo && room > 0) {
                        const int i = __ffsll((long long)todo) - 1;
                        kept_mask |= 1ull << i;
                        const box_t bi = {lane_value(bx.x1, i), lane_value(bx.y1, i), lane_value(bx.x2, i), lane_value(bx.y2, i)};
                        const float ai = lane_value(area, i);
                        todo &= ~(suppressed_lanes(bi, ai, bx, area, TH, lane > i) | (1ull << i));
                        --room;
                    }
                    keptm[j] = kept_mask;
                    if ((kept_mask >> lane) & 1ull) s_keep[start + p] = 1;
                    kept += __popcll(kept_mask);
                }
            }
            __syncthreads();
            // the kept rows join the image's list (any order: the last workgroup sorts what it needs)
            for (int base = 0; base < ns; base += 1024) {
                const int i = base + threadIdx.x;
                const bool k = i < ns && s_keep[i] != 0;
                const u64 bm = __ballot(k);
                if (lane == 0) wsum[wave] = __popcll(bm);
                __syncthreads();
                int before = 0, all = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) { const int cw = wsum[w]; if (w < wave) before += cw; all += cw; }
                if (threadIdx.x == 0) s_base = all ? atomicAdd(&kept_count[b], all) : 0;
                __syncthreads();
                const int at = s_base + before + __popcll(bm & lanemask_lt());
                if (k && at < m_cap) {
                    const u64 ck = s_key[SIDX(i)];
                    const uint32_t cls = (uint32_t)(ck >> 55), anchor = (uint32_t)ck & ((1u << GRP_ANCHOR_BITS) - 1u);
                    const float4 v = s_box[i];
                    float* __restrict__ o = kept_rows + (size_t)at * 6;
                    o[0] = __uint_as_float((uint32_t)(ck >> GRP_ANCHOR_BITS));
                    o[1] = __uint_as_float(anchor * ncd.d + cls);
                    o[2] = v.x; o[3] = v.y; o[4] = v.z; o[5] = v.w;
                }
                __syncthreads();
            }
        }
    }
    if (threadIdx.x == 0) {
        if (left) atomicOr(&left_flag[b], 1);
        if (g == 0) truncated_out[b] = truncated ? 1 : 0;
    }
}

// d  One workgroup per image: the first max_det kept rows in score order, from the list the class groups appended to.
__global__ void __launch_bounds__(1024) k_nms_groups_final(dh_fastdiv ncd, const float* __restrict__ kept_all, int m_cap,
                                                           const int* __restrict__ counts, const int* __restrict__ kept_count,
                                                           const int* __restrict__ truncated_in, const int* __restrict__ left_flag,
                                                           int max_det, float* __restrict__ out, int* __restrict__ cursor,
                                                           int2* __restrict__ out_span, int* __restrict__ cand_count,
                                                           int* __restrict__ needs_full, int out_rows)
{
    __shared__ u64 s_key[SORT_LDS_SLOTS(GRP_MAX)];    // key i at SIDX(i)
    __shared__ int hist[GRP_BINS];
    __shared__ int wsum[16], sh[4], s_fill, s_base;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const float* __restrict__ kept_rows = kept_all + (size_t)b * m_cap * 6;
    if (threadIdx.x == 0) cand_count[b] = counts[b];
    const int is_left = left_flag[b];
    const bool truncated = truncated_in[b] != 0;
    int K = kept_count[b];
    if (K > m_cap) K = m_cap;
    if (K > GRP_MAX) K = GRP_MAX;
    if (is_left || K <= 0) {                                            // workgroup-uniform
        if (threadIdx.x == 0) { out_span[b] = make_int2(0, 0); needs_full[b] = is_left ? 2 : (truncated ? 1 : 0); }
        return;
    }
    const int total = K < max_det ? K : max_det;
    // this thread's kept rows: {score order, anchor*nc + class} of row j, GRP_MAX / 1024 of them at most
    u64 mykey[GRP_MAX / 1024];
#pragma unroll
    for (int u = 0; u < GRP_MAX / 1024; ++u) {
        const int j = threadIdx.x + 1024 * u;
        mykey[u] = KEY_SENTINEL;
        if (j < K) {
            const float* __restrict__ r = kept_rows + (size_t)j * 6;
            mykey[u] = ((u64)__float_as_uint(r[0]) << 32) | __float_as_uint(r[1]);
        }
    }
    // candidates for the output: the best bins of the kept rows' scores that hold at least `total` rows
    int cutK = GRP_BINS - 1, nsel = K;
    if (K > GRP_FINAL) {
        for (int i = threadIdx.x; i < GRP_BINS; i += 1024) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < GRP_MAX / 1024; ++u)
            if (threadIdx.x + 1024 * u < K) atomicAdd(&hist[score_bin((uint32_t)(mykey[u] >> 32))], 1);
        __syncthreads();
        cut_bin(hist, total, wsum, sh, cutK, nsel);
    }
    if (threadIdx.x == 0) s_fill = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GRP_MAX / 1024; ++u) {
        const bool take = threadIdx.x + 1024 * u < K && (int)score_bin((uint32_t)(mykey[u] >> 32)) <= cutK;
        const u64 bm = __ballot(take);
        if (bm == 0) continue;                                          // wave-uniform
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&s_fill, __popcll(bm));
        slot = __builtin_amdgcn_readfirstlane(slot) + __popcll(bm & lanemask_lt());
        if (take && slot < GRP_MAX) s_key[SIDX(slot)] = mykey[u];
    }
    __syncthreads();
    nsel = s_fill < GRP_MAX ? s_fill : GRP_MAX;
    {
        const uint32_t npad = pow2_ceil((uint32_t)(nsel < 2 ? 2 : nsel));
        for (uint32_t i = nsel + threadIdx.x; i < npad; i += 1024) s_key[SIDX(i)] = KEY_SENTINEL;
        __syncthreads();
        if (npad <= 1024) bitonic_small(s_key, npad); else bitonic_lds(s_key, npad, 0, 2, npad);
    }
    if (threadIdx.x == 0) {
        needs_full[b] = (truncated && K < max_det) ? 1 : 0;
        const int first = atomicAdd(cursor, total);
        s_base = first;
        out_span[b] = make_int2(first, total);
    }
    __syncthreads();
    const int first = s_base;
    // every kept row looks its rank up in the sorted keys (they are all different) and, if it is among the first
    // `total`, writes its row there
#pragma unroll
    for (int u = 0; u < GRP_MAX / 1024; ++u) {
        const int j = threadIdx.x + 1024 * u;
        const u64 key = mykey[u];
        if (j >= K || (int)score_bin((uint32_t)(key >> 32)) > cutK) continue;
        int lo = 0, hi = nsel;                      // first position with s_key >= key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key[SIDX(mid)] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < total && (long long)first + lo < out_rows) {
            const float* __restrict__ r = kept_rows + (size_t)j * 6;
            const uint32_t ord = (uint32_t)key;
            const uint32_t anchor = dh_div(ord, ncd);
            float* __restrict__ o = out + ((size_t)first + lo) * 6;
            o[0] = r[2]; o[1] = r[3]; o[2] = r[4]; o[3] = r[5];
            o[4] = key_score((uint32_t)(key >> 32));
            o[5] = (float)(ord - anchor * ncd.d);
        }
    }
}

// ------------------------------------------------------------------------------------------ grouped rows (CenterNet)
// Ready-made detection rows [x1,y1,x2,y2,conf,cls] of several images (image b = count[b] rows from row first[b])
// instead of a prediction tensor: keys from the row's own score (or none: input order), gather by row index, and a collect
// that walks the (class, rank) order so the output is grouped by ascending class.
__global__ void __launch_bounds__(256) k_rows_keys(const float* __restrict__ rows, const int* __restrict__ first,
                                                   const int* __restrict__ count, int max_rows, int use_score,
                                                   u64* __restrict__ keys, int cap_pad, int* __restrict__ counts)
{
    const int b = blockIdx.y;
    const int f = first[b];
    int n = count[b];
    n = n < 0 ? 0 : (n > max_rows ? max_rows : n);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) counts[b] = n;
    if (j < n) keys[(size_t)b * cap_pad + j] = ((u64)(use_score ? score_key(rows[(size_t)(f + j) * 6 + 4]) : 0u) << 32) | (uint32_t)j;
}

__global__ void __launch_bounds__(256) k_rows_gather(const float* __restrict__ rows, const int* __restrict__ first,
                                                     const u64* __restrict__ keys, int cap_pad, const int* __restrict__ m_arr,
                                                     float* __restrict__ det, u64* __restrict__ keys2, int m_pad, int m_cap,
                                                     int nc)
{
    const int b = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m_arr[b]) return;
    const uint32_t j = (uint32_t)keys[(size_t)b * cap_pad + r];
    const float* __restrict__ p = rows + (size_t)(first[b] + j) * 6;
    float* __restrict__ d = det + ((size_t)b * m_cap + r) * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) d[q] = p[q];
    int cls = (int)p[5];
    cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);      // the caller guarantees 0 <= class < nc; never index outside the table
    keys2[(size_t)b * m_pad + r] = ((u64)(uint32_t)cls << 32) | (uint32_t)r;
}

__global__ void __launch_bounds__(1024) k_rows_collect(const float* __restrict__ det, const unsigned char* __restrict__ keep,
                                                       const u64* __restrict__ keys2, int m_pad, int m_cap,
                                                       const int* __restrict__ m_arr, const int* __restrict__ out_first, int use_keep,
                                                       float* __restrict__ out, int* __restrict__ out_count)
{
    const int b = blockIdx.x;
    const int m = m_arr[b];
    __shared__ int wcount[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* __restrict__ o_img = out + (size_t)out_first[b] * 6;
    int done = 0;
    for (int base = 0; base < m; base += 1024) {
        const int p = base + threadIdx.x;
        int rank = 0;
        bool k = false;
        if (p < m) {
            rank = (int)(uint32_t)keys2[(size_t)b * m_pad + p];
            k = !use_keep || keep[(size_t)b * m_cap + rank] != 0;
        }
        const u64 bm = __ballot(k);
        if (lane == 0) wcount[wave] = __popcll(bm);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = wcount[w]; if (w < wave) before += c; all += c; }
        if (k) {
            const float* __restrict__ d = det + ((size_t)b * m_cap + rank) * 6;
            float* __restrict__ o = o_img + (size_t)(done + before + __popcll(bm & lanemask_lt())) * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = d[q];
        }
        done += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_count[b] = done;
}

// plain nms(): keys from scores, det rows from boxes
__global__ void __launch_bounds__(256) k_nms_plain_keys(const float* __restrict__ scores, int n, u64* __restrict__ keys,
                                                        int* __restrict__ counts, int* __restrict__ m_arr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((u64)score_key(scores[i]) << 32) | (uint32_t)i;
    if (i == 0) { counts[0] = n; m_arr[0] = n; }
}

__global__ void __launch_bounds__(256) k_nms_plain_gather(const float* __restrict__ boxes, const u64* __restrict__ keys,
                                                          int n, float* __restrict__ det)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u64 key = keys[r];
    const float* __restrict__ p = boxes + (size_t)(uint32_t)key * 4;
    float* __restrict__ d = det + (size_t)r * 6;
    d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3];
    d[4] = key_score((uint32_t)(key >> 32));
    d[5] = 0.0f;
}

struct nms_layout {
    int cap_pad, m_cap, m_pad;
    size_t off_counts, off_m, off_trunc, off_flag, off_range, off_cursor, off_seg, off_keys, off_top, off_keys2, off_det, off_keep, off_spill, total;
    int top_pad;         // keys per image in the select-sort output (0: selection not used)
    size_t zero_bytes;   // [0, zero_bytes) = counts, ranges, cursor, segment table, keep flags: cleared by one memset
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

uint32_t pow2_ceil_host(uint32_t n)
{
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// smallest selection target >= k_select that leaves room for 8*max_det rows; 0 when that does not fit the LDS sorter
int select_target(int k_select, int max_det, int cap, int max_nms)
{
    if (k_select <= 0) return 0;
    long k = k_select;
    while (k < 8l * max_det) k = 2 * (k + 256) - 256;
    if (k > SORT_LDS_KEYS - 256) return 0;
    const int all = cap < max_nms ? cap : max_nms;
    return k < all ? (int)k : 0;          // nothing to cut when every candidate fits anyway
}

nms_layout make_layout(int bs, int cap, int max_nms, int nc, bool with_top = true)
{
    nms_layout L;
    L.cap_pad = (int)pow2_ceil_host((uint32_t)(cap < 2 ? 2 : cap));
    L.m_cap = cap < max_nms ? cap : max_nms;
    if (L.m_cap < 1) L.m_cap = 1;
    L.m_pad = (int)pow2_ceil_host((uint32_t)(L.m_cap < 2 ? 2 : L.m_cap));
    size_t o = 0;
    L.off_counts = o; o = align_up(o + sizeof(int) * bs, 256);
    L.off_m = o; o = align_up(o + sizeof(int) * bs, 256);
    L.off_trunc = o; o = align_up(o + sizeof(int) * bs, 256);
    L.off_flag = o; o = align_up(o + sizeof(int) * bs, 256);
    L.off_range = o; o = align_up(o + sizeof(img_range) * bs, 256);
    L.off_cursor = o; o = align_up(o + sizeof(int), 256);
    L.off_seg = o; o = align_up(o + sizeof(int2) * (size_t)bs * nc, 256);
    L.off_keep = o; o = align_up(o + (size_t)bs * L.m_cap, 256);
    L.zero_bytes = o;
    L.off_keys = o; o = align_up(o + sizeof(u64) * (size_t)bs * L.cap_pad, 256);
    L.top_pad = with_top ? SORT_LDS_KEYS : 0;
    L.off_top = o; o = align_up(o + sizeof(u64) * (size_t)bs * L.top_pad, 256);
    L.off_keys2 = o; o = align_up(o + sizeof(u64) * (size_t)bs * L.m_pad, 256);
    L.off_det = o; o = align_up(o + sizeof(float) * 6 * (size_t)bs * L.m_cap, 256);
    L.off_spill = o; o = align_up(o + sizeof(float4) * (size_t)bs * L.m_cap, 256);
    L.total = o;
    return L;
}

float round_down_to_float(double t)
{
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
}

}  // namespace

extern "C" size_t dh_nms_workspace(int bs, int nc, int cap, int max_nms)
{
    if (bs <= 0 || nc <= 0 || cap <= 0 || max_nms <= 0) return 0;
    return make_layout(bs, cap, max_nms, nc).total;
}

namespace {

int nms_run(const nms_source& S, bool raw, bool half, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
            int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
            int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
            size_t workspace_bytes, void* stream)
{
    if (bs < 0 || bs > 65535 || A < 1 || nc < 1 || max_det < 1 || max_nms < 1 || cap < 1) return DH_ERR_ARG;
    if ((long long)A * nc >= (1ll << 32) || (long long)bs * max_det >= (1ll << 31)) return DH_ERR_ARG;
    if (bs == 0) return DH_OK;
    if (!out || !out_span || !cand_count || !needs_full || !workspace) return DH_ERR_ARG;
    const nms_layout L = make_layout(bs, cap, max_nms, nc);
    // k_select = -2: everything after the candidate search in one launch (k_nms_groups) where its tables fit; otherwise
    // the same kind of score prefix goes through the separate launches
    const bool chain = k_select == -2 && !agnostic && max_wh >= 0.0 && nc <= GRP_NC && max_det <= GRP_KEPT &&
                       8l * max_det <= GRP_TARGET && A <= (1 << GRP_ANCHOR_BITS);
    if (k_select == -2 && !chain) k_select = GRP_MAX - 256;
    if (k_select < -2) return DH_ERR_ARG;
    const int k_target = select_target(k_select, max_det, cap, max_nms);
    if (workspace_bytes < L.total) return DH_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    int* counts = (int*)(ws + L.off_counts);
    int* m_arr = (int*)(ws + L.off_m);
    int* truncated = (int*)(ws + L.off_trunc);
    img_range* range = (img_range*)(ws + L.off_range);
    int2* seg = (int2*)(ws + L.off_seg);
    unsigned char* keep = (unsigned char*)(ws + L.off_keep);
    u64* keys = (u64*)(ws + L.off_keys);
    u64* keys2 = (u64*)(ws + L.off_keys2);
    float* det = (float*)(ws + L.off_det);
    float4* spill = (float4*)(ws + L.off_spill);
    hipStream_t st = (hipStream_t)stream;
    const float conf = (float)conf_thres;
    const float thr = round_down_to_float(iou_thres);

    int* cursor = (int*)(ws + L.off_cursor);
    if (dh_fill_async(ws, 0, L.zero_bytes, st) != DH_OK) return DH_ERR_LAUNCH;   // every counter's initial state is 0
    // Multi-label at a validation-style threshold: most rows are candidates, so the tensor (decoded, or the raw maps) is
    // streamed once instead of being visited candidate row by candidate row.
    bool streamed = false;
    if (S.conv_na) {          // convolution layout: its own candidate search, everything after it is shared
        conv_blocks CB;
        uint32_t cgx = 0;
        for (int l = 0; l < S.nl; ++l) {
            CB.blocks[l] = (uint32_t)S.conv_na * ((S.lvl[l].plane.d + 255) / 256);
            cgx += CB.blocks[l];
        }
#define DH_CONV(M, T, H) hipLaunchKernelGGL((k_nms_prefilter_conv<M, T, H>), dim3(cgx, bs), dim3(256), 0, st, S, CB, nc, conf, class_keep, cap, keys, L.cap_pad, counts)
        if (half && S.half_arith) { if (multi_label) DH_CONV(true, _Float16, true); else DH_CONV(false, _Float16, true); }
        else if (half) { if (multi_label) DH_CONV(true, _Float16, false); else DH_CONV(false, _Float16, false); }
        else { if (multi_label) DH_CONV(true, float, false); else DH_CONV(false, float, false); }
#undef DH_CONV
        streamed = true;
    }
    if (!streamed && multi_label && conf_thres < 0.05 && nc + 5 >= 7) {
        stream_params SP;
        SP.no = dh_make_fastdiv((uint32_t)(nc + 5));
        bool ok = true;
        uint32_t sgx = 0;
        if (raw) {
            SP.nseg = S.nl;
            for (int l = 0; l < S.nl; ++l) {
                SP.seg[l].src = S.lvl[l].src; SP.seg[l].elems = S.lvl[l].elems; SP.seg[l].row_first = S.lvl[l].row_first;
            }
        } else {
            SP.nseg = 1;
            SP.seg[0].src = S.pred; SP.seg[0].row_first = 0;
            ok = (unsigned long long)A * (nc + 5) < (1ull << 31);
            SP.seg[0].elems = ok ? (uint32_t)A * (uint32_t)(nc + 5) : 0;
        }
        SP.tiles = 0;
        for (int l = 0; ok && l < SP.nseg; ++l) {
            ok = (SP.seg[l].elems % 4 == 0) && (half ? (reinterpret_cast<uintptr_t>(SP.seg[l].src) & 7u) == 0 : dh_aligned16(SP.seg[l].src));
            SP.seg[l].tiles = (SP.seg[l].elems / 4 + STREAM_TILE - 1) / STREAM_TILE;
            SP.tiles += SP.seg[l].tiles;
        }
        // as many workgroups as the chip holds at once, shared out evenly over the images; none shorter than STREAM_MIN_TILES
        const int cus = dh_cu_count();
        uint32_t per_image = (uint32_t)((cus > 0 ? cus : 256) * STREAM_WG_PER_CU * STREAM_ROUNDS / bs);
        if (per_image < 1) per_image = 1;
        const uint32_t most = (SP.tiles + STREAM_MIN_TILES - 1) / STREAM_MIN_TILES;
        if (per_image > most) per_image = most;
        ok = ok && SP.tiles > 0 && SP.tiles < (1u << 30);
        sgx = per_image;
        if (ok) {
            if (raw && half && S.half_arith)
                hipLaunchKernelGGL((k_nms_prefilter_stream<true, _Float16, true>), dim3(sgx, bs), dim3(256), 0, st, SP, obj_plane,
                                   A, nc, conf, class_keep, cap, keys, L.cap_pad, counts);
            else if (raw && half) hipLaunchKernelGGL((k_nms_prefilter_stream<true, _Float16>), dim3(sgx, bs), dim3(256), 0, st, SP, obj_plane,
                                                A, nc, conf, class_keep, cap, keys, L.cap_pad, counts);
            else if (raw) hipLaunchKernelGGL(k_nms_prefilter_stream<true>, dim3(sgx, bs), dim3(256), 0, st, SP, obj_plane, A, nc, conf,
                                             class_keep, cap, keys, L.cap_pad, counts);
            else if (half) hipLaunchKernelGGL((k_nms_prefilter_stream<false, _Float16, true>), dim3(sgx, bs), dim3(256), 0, st, SP,
                                              obj_plane, A, nc, conf, class_keep, cap, keys, L.cap_pad, counts);
            else hipLaunchKernelGGL(k_nms_prefilter_stream<false>, dim3(sgx, bs), dim3(256), 0, st, SP, obj_plane, A, nc, conf,
                                    class_keep, cap, keys, L.cap_pad, counts);
            streamed = true;
        }
    }
    int gx = (A + 255) / 256;                       // one 64-row group per wave, unless that makes > 8192 blocks:
    const int gx_cap = 8192 / bs > 1 ? 8192 / bs : 1;
    if (gx > gx_cap) {                              // then every workgroup takes the same number of groups (one fewer for the
        const int rounds = (gx + gx_cap - 1) / gx_cap;      // last few), instead of some taking two and the rest one
        gx = (gx + rounds - 1) / rounds;
    }
    // dense single-label path: 4 waves x DENSE_ROWS rows x no floats of LDS, when a tile fits the per-lane registers
    int dense = (!multi_label && nc + 5 <= 64 * DENSE_REGS / DENSE_ROWS) ? 1 : 0;
    if (dense && !raw && !half && dh_aligned16(S.pred) && ((long long)A * (nc + 5)) % 4 == 0) dense = 2;      // tiles as 16-byte vectors
    const size_t dense_lds = dense ? sizeof(float) * 4 * DENSE_ROWS * (nc + 5) : 0;
#define DH_PREFILTER(M, N, R, T, H) hipLaunchKernelGGL((k_nms_prefilter<M, N, R, T, H>), dim3(gx, bs), dim3(256), (M) ? 0 : dense_lds, st, S, obj_plane, A, nc, conf, class_keep, cap, keys, L.cap_pad, counts, dense)
#define DH_PREFILTER_N(M, R, T, H) do { if (nc <= 64) DH_PREFILTER(M, 1, R, T, H); else if (nc <= 128) DH_PREFILTER(M, 2, R, T, H); else DH_PREFILTER(M, 0, R, T, H); } while (0)
    if (streamed) { /* candidates already emitted */ }
    else if (raw && half && S.half_arith) { if (multi_label) DH_PREFILTER_N(true, true, _Float16, true); else DH_PREFILTER_N(false, true, _Float16, true); }
    else if (raw && half) { if (multi_label) DH_PREFILTER_N(true, true, _Float16, false); else DH_PREFILTER_N(false, true, _Float16, false); }
    else if (half) { if (multi_label) DH_PREFILTER_N(true, false, _Float16, true); else DH_PREFILTER_N(false, false, _Float16, true); }
    else if (multi_label) { if (raw) DH_PREFILTER_N(true, true, float, false); else DH_PREFILTER_N(true, false, float, false); }
    else { if (raw) DH_PREFILTER_N(false, true, float, false); else DH_PREFILTER_N(false, false, float, false); }
#undef DH_PREFILTER_N
#undef DH_PREFILTER
    if (chain) {
        // G workgroups per image, each taking the classes c % G == g: enough of them to fill the chip
        int G = 256 / bs;
        G = G < 1 ? 1 : (G > 16 ? 16 : G);
        if (G > nc) G = nc;
        int* kept_count = m_arr;                                        // zero after the fill above
        int* left_flag = (int*)(ws + L.off_flag);
#define DH_GROUPS(R, T) hipLaunchKernelGGL((k_nms_groups<R, T>), dim3(G, bs), dim3(1024), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), keys, L.cap_pad, counts, cap, max_nms, (float)max_wh, thr, max_det, det, L.m_cap, kept_count, truncated, left_flag)
        if (raw && half) DH_GROUPS(true, _Float16); else if (raw) DH_GROUPS(true, float); else if (half) DH_GROUPS(false, _Float16);
        else DH_GROUPS(false, float);
#undef DH_GROUPS
        hipLaunchKernelGGL(k_nms_groups_final, dim3(bs), dim3(1024), 0, st, dh_make_fastdiv((uint32_t)nc), det, L.m_cap, counts, kept_count,
                           truncated, left_flag, max_det, out, cursor, (int2*)out_span, cand_count, needs_full, bs * max_det);
        return dh_launch_status();
    }
    if (k_select == -1) {      // one workgroup per image does the rest; images with too many candidates set needs_full
        if (max_det > SMALL_MAX) return DH_ERR_ARG;
        const float off = agnostic ? 0.0f : (float)max_wh;
#define DH_SMALL(R, T) hipLaunchKernelGGL((k_nms_small<R, T>), dim3(bs), dim3(1024), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), keys, L.cap_pad, counts, cap, agnostic, off, thr, max_det, out, cursor, (int2*)out_span, cand_count, needs_full, bs * max_det)
        if (raw && half) DH_SMALL(true, _Float16); else if (raw) DH_SMALL(true, float); else if (half) DH_SMALL(false, _Float16);
        else DH_SMALL(false, float);
#undef DH_SMALL
        return dh_launch_status();
    }
    // score order: either the best k_target rows per image (select + sort in LDS) or all of them
    const u64* ranked = keys;
    int ranked_pad = L.cap_pad;
    if (k_target > 0) {
        ranked = (u64*)(ws + L.off_top);
        ranked_pad = L.top_pad;
        hipLaunchKernelGGL(k_select_sort, dim3(bs), dim3(SORT_THREADS), 0, st, keys, counts, cap, L.cap_pad, k_target, max_nms,
                           (u64*)(ws + L.off_top), L.top_pad, m_arr, truncated);
    } else {
        hipLaunchKernelGGL(k_sort_keys, dim3(bs), dim3(SORT_THREADS), 0, st, keys, counts, cap, L.cap_pad, (int2*)nullptr, 0,
                           m_arr, max_nms);
    }
    const int m_rows = k_target > 0 ? SORT_LDS_KEYS : L.m_cap;         // upper bound of m_arr[]
    const dim3 ggrid((m_rows + 255) / 256, bs);
    if (raw && half)
        hipLaunchKernelGGL((k_nms_gather<true, _Float16>), ggrid, dim3(256), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), ranked,
                           ranked_pad, m_arr, det, keys2, L.m_pad, L.m_cap, range);
    else if (raw)
        hipLaunchKernelGGL(k_nms_gather<true>, ggrid, dim3(256), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), ranked, ranked_pad,
                           m_arr, det, keys2, L.m_pad, L.m_cap, range);
    else if (half)
        hipLaunchKernelGGL((k_nms_gather<false, _Float16>), ggrid, dim3(256), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), ranked,
                           ranked_pad, m_arr, det, keys2, L.m_pad, L.m_cap, range);
    else
        hipLaunchKernelGGL(k_nms_gather<false>, ggrid, dim3(256), 0, st, S, A, dh_make_fastdiv((uint32_t)nc), ranked, ranked_pad,
                           m_arr, det, keys2, L.m_pad, L.m_cap, range);
    if (!agnostic) {
        hipLaunchKernelGGL(k_sort_keys, dim3(bs), dim3(SORT_THREADS), 0, st, keys2, m_arr, L.m_cap, L.m_pad, seg, nc,
                           (int*)nullptr, 0);
        hipLaunchKernelGGL(k_nms_greedy<4>, dim3(nc, bs), dim3(256), 0, st, det, keys2, L.m_pad, L.m_cap, m_arr, range, seg, 0,
                           (float)max_wh, thr, max_det, keep, spill);
    } else {
        hipLaunchKernelGGL(k_nms_greedy<16>, dim3(1, bs), dim3(1024), 0, st, det, keys2, L.m_pad, L.m_cap, m_arr, range, seg, 1,
                           (float)max_wh, thr, max_det, keep, spill);
    }
    hipLaunchKernelGGL(k_nms_collect, dim3(bs), dim3(1024), 0, st, det, keep, ranked, ranked_pad, L.m_cap, counts, m_arr,
                       k_target > 0 ? truncated : (const int*)nullptr, max_det, out, (long long*)nullptr, (int*)nullptr, cursor,
                       (int2*)out_span, cand_count, needs_full, bs * max_det);
    return dh_launch_status();
}

}  // namespace

extern "C" int dh_nms_batched(const float* pred, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
                              int multi_label, int agnostic, const unsigned char* class_keep, int max_det,
                              int max_nms, double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                              int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    if (bs > 0 && !pred) return DH_ERR_ARG;
    nms_source S;
    S.pred = pred;
    S.nl = 0;
    S.no = nc + 5;
    S.conv_na = 0;
    S.corners = 0;
    S.half_arith = 0;
    return nms_run(S, false, false, obj_plane, bs, A, nc, conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det, max_nms,
                   max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------ half-precision predictions
// A decoded tensor in fp16 is what the reference's NMS sees behind a half-precision model (val.py:129-130,185).  Its
// arithmetic is then fp16 up to the point where the detection matrix is assembled: conf = obj * cls is rounded to fp16
// (general.py:691), the corners are formed in fp16 (general.py:694,553-560), and torch.cat with the float class column
// (general.py:699,702) widens both; everything after that is fp32.  An image with apriori labels is widened earlier, by
// the cat with the float label rows (general.py:678-684), and has the fp32 arithmetic throughout.  This kernel does the
// fp16 part and leaves an fp32 tensor the fp32 pipeline turns into exactly the reference's rows: corners in channels 0..3,
// conf in the class channels, and in channel 4 the outcome of the objectness test (1 or 0) -- so that the pipeline's own
// "obj > conf" keeps what passed, its "obj * cls" returns conf unchanged, and the corners are taken as they are
// (nms_source::corners).  Thresholds are compared in fp32, as the reference's device kernels compare an fp16 tensor with
// a Python scalar.
// One thread: VEC consecutive elements of one image.  VEC = 8 (4): one 16-byte (8-byte) load, two (one) 16-byte stores; needs
// A * no % VEC == 0 and no >= VEC + 4, so that a vector holds channels 0..4 of at most one row (`row`, or `row + 1` when it
// runs over the row's end).
// Every element is first treated as a class score (one multiply, one rounding); the few vectors that hold box or
// objectness channels patch them afterwards.  plane (optional): the objectness outcomes once more, compact, (bs, A).
template <int VEC>
__global__ void __launch_bounds__(256) k_nms_prepare_f16(const _Float16* __restrict__ pred, uint32_t per_image, const dh_fastdiv nod,
                                                         float conf, const unsigned char* __restrict__ image_fp32,
                                                         float* __restrict__ out, float* __restrict__ plane)
{
    const int b = blockIdx.y;
    const uint32_t no = nod.d;
    const uint32_t nvec = per_image / VEC;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (VEC != 8 && v >= nvec) return;
    if (VEC == 8 && v - (threadIdx.x & 63) >= nvec) return;                 // whole waves only: see the stores
    const _Float16* __restrict__ p = pred + (size_t)b * per_image;         // this image's rows
    float* __restrict__ o = out + (size_t)b * per_image + (size_t)v * VEC;
    const bool wide = image_fp32 != nullptr && image_fp32[b] != 0;
    const uint32_t e = (v < nvec ? v : nvec - 1) * VEC;                     // a lane past the end repeats the last vector
    const uint32_t row = dh_div(e, nod), c0 = e - row * no;
    _Float16 x[VEC];
    typedef _Float16 dh_h8 __attribute__((ext_vector_type(8)));
    if (VEC == 8) *reinterpret_cast<dh_h8*>(x) = __builtin_nontemporal_load(reinterpret_cast<const dh_h8*>(p + e));
    else if (VEC == 4) *reinterpret_cast<dh_h4*>(x) = __builtin_nontemporal_load(reinterpret_cast<const dh_h4*>(p + e));
    else x[0] = p[e];
    const bool over = c0 + VEC > no;                                        // runs over into row + 1 (of the same image)
    const float obj0 = (float)p[(size_t)row * no + 4];
    const float obj1 = over ? (float)p[(size_t)(row + 1) * no + 4] : obj0;
    // The box of the row whose channels 0..4 this vector may hold (row, or row + 1 when the vector starts beyond channel 4)
    // is fetched by every lane, needed or not, together with the loads above: inside the branch below the four loads would
    // be a second round trip for nearly every wave (a wave's 512 elements always contain some row's box).
    const bool next = c0 >= 5;
    const uint32_t rbox = min(row + (next ? 1u : 0u), per_image / no - 1);
    const _Float16* __restrict__ pr = p + (size_t)rbox * no;
    const float bx = (float)pr[0], by = (float)pr[1], bw = (float)pr[2], bh = (float)pr[3];
    float y[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float val = (float)x[k] * (c0 + k >= no ? obj1 : obj0);      // general.py:691 (exact in fp32: 11 x 11 bits)
        y[k] = wide ? val : dh_round_f16(val);
    }
    if (c0 < 5 || over) {
        const uint32_t r = row + (next ? 1u : 0u);
        const int k0 = next ? (int)(no - c0) : -(int)c0;                    // element of this vector where channel 0 of row r sits
        const float obj = next ? obj1 : obj0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = k - k0;
            if (c >= 0 && c < 4) {                                          // this element's corner along axis c & 1 (general.py:556-559)
                const float ctr = (c & 1) ? by : bx, sz = (c & 1) ? bh : bw;
                const float half_size = wide ? sz / 2 : dh_round_f16(sz / 2);
                const float corner = (c & 2) ? ctr + half_size : ctr - half_size;
                y[k] = wide ? corner : dh_round_f16(corner);
            } else if (c == 4) {
                y[k] = obj > conf ? 1.0f : 0.0f;                            // general.py:656,675
                if (plane) plane[(size_t)b * (per_image / no) + r] = y[k];
            }
        }
    }
    if (VEC == 8) {
        // A lane holds 8 consecutive floats, i.e. two 16-byte pieces 32 bytes apart from its neighbour's: stored as they are,
        // every store instruction would fill half of each 64-byte sector.  The wave's 128 pieces are redistributed instead:
        // store q writes pieces 64q .. 64q+63, lane i piece 64q + i, which lane 32q + i/2 holds in its half i & 1.  Every lane
        // of the wave takes part (lanes past the end hold a repeat of the last vector and store nothing).
        const int lane = threadIdx.x & 63;
        float* __restrict__ wave_out = out + (size_t)b * per_image + (size_t)(v - lane) * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int src = 32 * q + (lane >> 1);
            float z[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = __shfl(y[k], src, 64), hi = __shfl(y[4 + k], src, 64);
                z[k] = (lane & 1) ? hi : lo;
            }
            const uint32_t piece = 64 * q + lane;                  // of the wave's 128
            if ((v - lane) * 2 + piece < nvec * 2) {
                const dh_f4 z4 = {z[0], z[1], z[2], z[3]};
                __builtin_nontemporal_store(z4, reinterpret_cast<dh_f4*>(wave_out) + piece);
            }
        }
    } else if (VEC == 4) {
        const dh_f4 y4 = {y[0], y[1], y[2], y[3]};
        __builtin_nontemporal_store(y4, reinterpret_cast<dh_f4*>(o));
    } else {
        o[0] = y[0];
    }
}

extern "C" int dh_nms_prepare_f16(const void* pred_f16, int bs, int A, int nc, double conf_thres, const unsigned char* image_fp32,
                                  float* out, float* obj_plane, void* stream)
{
    if (bs < 0 || bs > 65535 || A < 0 || nc < 1) return DH_ERR_ARG;
    if (bs == 0 || A == 0) return DH_OK;
    if (!pred_f16 || !out) return DH_ERR_ARG;
    const unsigned long long per_image = (unsigned long long)A * (nc + 5);
    if (per_image >= (1ull << 31)) return DH_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(pred_f16) & 1u) return DH_ERR_ALIGN;
    const int no = nc + 5;
    int vec = 1;
    if (dh_aligned16(out)) {
        if (no >= 12 && per_image % 8 == 0 && (reinterpret_cast<uintptr_t>(pred_f16) & 15u) == 0) vec = 8;
        else if (no >= 8 && per_image % 4 == 0 && (reinterpret_cast<uintptr_t>(pred_f16) & 7u) == 0) vec = 4;
    }
    const dh_fastdiv nod = dh_make_fastdiv((uint32_t)no);
    const uint32_t nvec = (uint32_t)(per_image / vec);
    const dim3 grid((nvec + 255) / 256, bs);
#define DH_PREPARE(V) hipLaunchKernelGGL(k_nms_prepare_f16<V>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)pred_f16, \
                                         (uint32_t)per_image, nod, (float)conf_thres, image_fp32, out, obj_plane)
    if (vec == 8) DH_PREPARE(8); else if (vec == 4) DH_PREPARE(4); else DH_PREPARE(1);
#undef DH_PREPARE
    return dh_launch_status();
}

extern "C" int dh_nms_batched_prepared(const float* pred, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
                                       int multi_label, int agnostic, const unsigned char* class_keep, int max_det,
                                       int max_nms, double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                       int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    if (bs > 0 && !pred) return DH_ERR_ARG;
    nms_source S;
    S.pred = pred;
    S.nl = 0;
    S.no = nc + 5;
    S.conv_na = 0;
    S.corners = 1;
    S.half_arith = 0;
    return nms_run(S, false, false, obj_plane, bs, A, nc, conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det, max_nms,
                   max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

// non_max_suppression on the fp16 decoded tensor itself: the rows of dh_nms_prepare_f16 + dh_nms_batched_prepared without the
// fp32 copy -- conf = obj * cls is rounded to fp16 where the candidates are looked for, the corners are formed in fp16 where
// the boxes are fetched.  Not for images with apriori labels (they are widened before that arithmetic: the prepared form).
extern "C" int dh_nms_batched_f16(const void* pred_f16, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
                                  int multi_label, int agnostic, const unsigned char* class_keep, int max_det,
                                  int max_nms, double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                  int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    if (bs > 0 && !pred_f16) return DH_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(pred_f16) & 1u) return DH_ERR_ALIGN;
    nms_source S;
    S.pred = reinterpret_cast<const float*>(pred_f16);
    S.nl = 0;
    S.no = nc + 5;
    S.conv_na = 0;
    S.corners = 0;
    S.half_arith = 1;
    return nms_run(S, false, true, obj_plane, bs, A, nc, conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det, max_nms,
                   max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

namespace {

// ------------------------------------------------------------------------------------------ merge-NMS
// general.py:725-731, a step the reference keeps behind the local switch `merge` (general.py:668).  For an image with
// 1 < n < 3000 candidates every output row's box becomes the score-weighted mean of all candidate boxes whose IoU with
// it exceeds the threshold (IoU of metrics.py:242-264 on the class-shifted boxes, weights = scores, the sum runs over the
// unshifted boxes), and rows that matched no candidate but themselves are dropped (`redundant`, general.py:666,730-731).
// One workgroup per image; one wave per output row walks the image's candidates (`det`, left in the workspace by the
// general NMS path in score order).  The rows are rewritten in place and closed up in order.
#define MERGE_MAX_DET 4096
__global__ void __launch_bounds__(1024) k_nms_merge(const float* __restrict__ det, int m_cap, const int* __restrict__ counts,
                                                    const int* __restrict__ m_arr, float off, float thr, int redundant,
                                                    float* __restrict__ out, int2* __restrict__ out_span)
{
    __shared__ unsigned char s_keep[MERGE_MAX_DET];
    __shared__ int wcount[16];
    const int b = blockIdx.x;
    const int n = counts[b], m = m_arr[b];
    if (!(n > 1 && n < 3000)) return;                              // general.py:725 (n: candidates before any truncation)
    const int2 span = out_span[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ cand = det + (size_t)b * m_cap * 6;
    for (int i = wave; i < span.y; i += 16) {
        float* __restrict__ o = out + ((size_t)span.x + i) * 6;
        const float shift = o[5] * off;                            // general.py:720
        const float x1 = o[0] + shift, y1 = o[1] + shift, x2 = o[2] + shift, y2 = o[3] + shift;    // general.py:721
        const float area = (x2 - x1) * (y2 - y1);
        float wsum = 0.0f, sx1 = 0.0f, sy1 = 0.0f, sx2 = 0.0f, sy2 = 0.0f;
        int hits = 0;
        for (int j = lane; j < m; j += 64) {
            const float* __restrict__ c = cand + (size_t)j * 6;
            const float u1 = c[0], v1 = c[1], u2 = c[2], v2 = c[3], score = c[4];
            const float cs = c[5] * off;
            const float a1 = u1 + cs, b1 = v1 + cs, a2 = u2 + cs, b2 = v2 + cs;
            const float carea = (a2 - a1) * (b2 - b1);
            const float inter = fmaxf(fminf(x2, a2) - fmaxf(x1, a1), 0.0f) * fmaxf(fminf(y2, b2) - fmaxf(y1, b1), 0.0f);
            const float iou = __fdiv_rn(inter, area + carea - inter);             // metrics.py:264
            if (iou > thr) {                                                      // general.py:727-728
                ++hits;
                wsum += score;
                sx1 += score * u1; sy1 += score * v1; sx2 += score * u2; sy2 += score * v2;
            }
        }
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) {
            hits += __shfl_xor(hits, k, 64);
            wsum += __shfl_xor(wsum, k, 64);
            sx1 += __shfl_xor(sx1, k, 64); sy1 += __shfl_xor(sy1, k, 64);
            sx2 += __shfl_xor(sx2, k, 64); sy2 += __shfl_xor(sy2, k, 64);
        }
        if (lane == 0) {
            o[0] = __fdiv_rn(sx1, wsum); o[1] = __fdiv_rn(sy1, wsum);             // general.py:729
            o[2] = __fdiv_rn(sx2, wsum); o[3] = __fdiv_rn(sy2, wsum);
            s_keep[i] = (!redundant || hits > 1) ? 1 : 0;                         // general.py:730-731
        }
    }
    __threadfence_block();
    __syncthreads();
    if (!redundant) return;
    int done = 0;
    for (int base = 0; base < span.y; base += 1024) {
        const int i = base + threadIdx.x;
        const bool k = i < span.y && s_keep[i] != 0;
        float row[6];
        if (k) {
#pragma unroll
            for (int q = 0; q < 6; ++q) row[q] = out[((size_t)span.x + i) * 6 + q];
        }
        const u64 bm = __ballot(k);
        if (lane == 0) wcount[wave] = __popcll(bm);
        __syncthreads();                      // every row of this chunk has been read before any is overwritten
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = wcount[w]; if (w < wave) before += c; all += c; }
        if (k) {
            const int pos = done + before + __popcll(bm & lanemask_lt());
#pragma unroll
            for (int q = 0; q < 6; ++q) out[((size_t)span.x + pos) * 6 + q] = row[q];
        }
        done += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_span[b] = make_int2(span.x, done);
}

}  // namespace

extern "C" int dh_nms_merge(float* out, int* out_span, int bs, int nc, int cap, int max_nms, int max_det, double iou_thres,
                            int agnostic, double max_wh, int redundant, void* workspace, size_t workspace_bytes, void* stream)
{
    if (bs < 0 || bs > 65535 || nc < 1 || cap < 1 || max_nms < 1 || max_det < 1 || max_det > MERGE_MAX_DET) return DH_ERR_ARG;
    if (bs == 0) return DH_OK;
    if (!out || !out_span || !workspace) return DH_ERR_ARG;
    const nms_layout L = make_layout(bs, cap, max_nms, nc);
    if (workspace_bytes < L.total) return DH_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    // box_iou(...) > iou_thres compares fp32 values with the threshold converted to fp32 (torch's type promotion of a
    // Python scalar against a float tensor), unlike the comparison in double inside the NMS proper
    hipLaunchKernelGGL(k_nms_merge, dim3(bs), dim3(1024), 0, (hipStream_t)stream, (const float*)(ws + L.off_det), L.m_cap,
                       (const int*)(ws + L.off_counts), (const int*)(ws + L.off_m), agnostic ? 0.0f : (float)max_wh,
                       (float)iou_thres, redundant ? 1 : 0, out, (int2*)out_span);
    return dh_launch_status();
}

namespace {

int decode_nms_run(const void* const* level_ptrs, bool half, bool conv, bool half_arith, const int* ny, const int* nx, int nl, int bs, int na, int no,
                   const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                   int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap, int k_select,
                   float* out, int* out_span, int* cand_count, int* needs_full, void* workspace, size_t workspace_bytes,
                   void* stream)
{
    if (nl < 1 || nl > DH_MAX_LEVELS || na < 1 || na > DH_MAX_ANCHORS || no < 6) return DH_ERR_ARG;
    if (!level_ptrs || !ny || !nx || !anchor_px || !strides) return DH_ERR_ARG;
    nms_source S;
    S.pred = nullptr;
    S.nl = nl;
    S.no = no;
    S.conv_na = conv ? na : 0;
    S.corners = 0;
    S.half_arith = half_arith ? 1 : 0;
    if (half_arith && !half) return DH_ERR_ARG;
    uint32_t per_image = 0, rows = 0;
    if (!dh_make_levels(reinterpret_cast<const float* const*>(level_ptrs), ny, nx, nl, na, no, anchor_px, strides, S.lvl, &per_image,
                        &rows, nullptr))
        return DH_ERR_ARG;
    for (int l = 0; half && l < nl; ++l)
        if (reinterpret_cast<uintptr_t>(level_ptrs[l]) & 1u) return DH_ERR_ALIGN;
    return nms_run(S, true, half, nullptr, bs, (int)rows, no - 5, conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det,
                   max_nms, max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

}  // namespace

extern "C" int dh_yolo_decode_nms(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                  const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                  int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                  double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                  int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(reinterpret_cast<const void* const*>(level_ptrs), false, false, false, ny, nx, nl, bs, na, no, anchor_px, strides,
                          conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out,
                          out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

extern "C" int dh_yolo_decode_nms_f16(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                      const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                      int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                      double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                      int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(level_ptrs, true, false, false, ny, nx, nl, bs, na, no, anchor_px, strides, conf_thres, iou_thres, multi_label,
                          agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace,
                          workspace_bytes, stream);
}

extern "C" int dh_yolo_decode_nms_half(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                       const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                       int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                       double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                       int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(level_ptrs, true, false, true, ny, nx, nl, bs, na, no, anchor_px, strides, conf_thres, iou_thres, multi_label,
                          agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out, out_span, cand_count, needs_full, workspace,
                          workspace_bytes, stream);
}

extern "C" int dh_yolo_conv_nms(const float* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(reinterpret_cast<const void* const*>(conv_ptrs), false, true, false, ny, nx, nl, bs, na, no, anchor_px, strides,
                          conf_thres, iou_thres, multi_label, agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out,
                          out_span, cand_count, needs_full, workspace, workspace_bytes, stream);
}

extern "C" int dh_yolo_conv_nms_f16(const void* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                    const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                    int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                    double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                    int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(conv_ptrs, true, true, false, ny, nx, nl, bs, na, no, anchor_px, strides, conf_thres, iou_thres, multi_label,
                          agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out, out_span, cand_count, needs_full,
                          workspace, workspace_bytes, stream);
}

extern "C" int dh_yolo_conv_nms_half(const void* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                                     const float* anchor_px, const float* strides, double conf_thres, double iou_thres,
                                     int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                                     double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count,
                                     int* needs_full, void* workspace, size_t workspace_bytes, void* stream)
{
    return decode_nms_run(conv_ptrs, true, true, true, ny, nx, nl, bs, na, no, anchor_px, strides, conf_thres, iou_thres, multi_label,
                          agnostic, class_keep, max_det, max_nms, max_wh, cap, k_select, out, out_span, cand_count, needs_full,
                          workspace, workspace_bytes, stream);
}

extern "C" size_t dh_group_nms_workspace(int n_img, int max_rows, int nc)
{
    if (n_img <= 0 || max_rows <= 0 || nc <= 0) return 0;
    return make_layout(n_img, max_rows, max_rows, nc, false).total;
}

extern "C" int dh_group_nms(const float* rows, const int* first, const int* count, const int* out_first, int n_img, int max_rows,
                            int nc, int need_nms, double nms_thres, float* out, int* out_count, void* workspace,
                            size_t workspace_bytes, void* stream)
{
    if (n_img < 0 || n_img > 65535 || max_rows < 0 || nc < 1) return DH_ERR_ARG;
    if (n_img == 0) return DH_OK;
    if (!first || !count || !out_first || !out_count || !workspace) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (max_rows == 0) return dh_fill_async(out_count, 0, sizeof(int) * n_img, st);
    if (!rows || !out) return DH_ERR_ARG;
    const nms_layout L = make_layout(n_img, max_rows, max_rows, nc, false);
    if (workspace_bytes < L.total) return DH_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    int* counts = (int*)(ws + L.off_counts);
    int* m_arr = (int*)(ws + L.off_m);
    img_range* range = (img_range*)(ws + L.off_range);
    int2* seg = (int2*)(ws + L.off_seg);
    unsigned char* keep = (unsigned char*)(ws + L.off_keep);
    u64* keys = (u64*)(ws + L.off_keys);
    u64* keys2 = (u64*)(ws + L.off_keys2);
    float* det = (float*)(ws + L.off_det);
    float4* spill = (float4*)(ws + L.off_spill);
    if (dh_fill_async(ws, 0, L.zero_bytes, st) != DH_OK) return DH_ERR_LAUNCH;
    const dim3 grid((max_rows + 255) / 256, n_img);
    hipLaunchKernelGGL(k_rows_keys, grid, dim3(256), 0, st, rows, first, count, max_rows, need_nms ? 1 : 0, keys, L.cap_pad, counts);
    hipLaunchKernelGGL(k_sort_keys, dim3(n_img), dim3(SORT_THREADS), 0, st, keys, counts, max_rows, L.cap_pad, (int2*)nullptr, 0,
                       m_arr, max_rows);
    hipLaunchKernelGGL(k_rows_gather, grid, dim3(256), 0, st, rows, first, keys, L.cap_pad, m_arr, det, keys2, L.m_pad, L.m_cap, nc);
    hipLaunchKernelGGL(k_sort_keys, dim3(n_img), dim3(SORT_THREADS), 0, st, keys2, m_arr, L.m_cap, L.m_pad, seg, nc, (int*)nullptr, 0);
    if (need_nms)
        hipLaunchKernelGGL(k_nms_greedy<4>, dim3(nc, n_img), dim3(256), 0, st, det, keys2, L.m_pad, L.m_cap, m_arr, range, seg, 0,
                           -1.0f, round_down_to_float(nms_thres), max_rows, keep, spill);
    hipLaunchKernelGGL(k_rows_collect, dim3(n_img), dim3(1024), 0, st, det, keep, keys2, L.m_pad, L.m_cap, m_arr, out_first,
                       need_nms ? 1 : 0, out, out_count);
    return dh_launch_status();
}

extern "C" size_t dh_nms_plain_workspace(int64_t n)
{
    if (n <= 0 || n >= (1ll << 31)) return 0;
    return make_layout(1, (int)n, (int)n, 1, false).total;
}

namespace {

int nms_plain_run(const float* boxes, const float* scores, int ni, double iou_thres, int64_t* keep_index, int* keep_count,
                  void* workspace, size_t workspace_bytes, hipStream_t st)
{
    const nms_layout L = make_layout(1, ni, ni, 1, false);
    if (workspace_bytes < L.total) return DH_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    int* counts = (int*)(ws + L.off_counts);
    int* m_arr = (int*)(ws + L.off_m);
    img_range* range = (img_range*)(ws + L.off_range);
    unsigned char* keep = (unsigned char*)(ws + L.off_keep);
    u64* keys = (u64*)(ws + L.off_keys);
    u64* keys2 = (u64*)(ws + L.off_keys2);
    float* det = (float*)(ws + L.off_det);
    float4* spill = (float4*)(ws + L.off_spill);
    if (dh_fill_async(ws, 0, L.zero_bytes, st) != DH_OK) return DH_ERR_LAUNCH;
    hipLaunchKernelGGL(k_nms_plain_keys, dim3((ni + 255) / 256), dim3(256), 0, st, scores, ni, keys, counts, m_arr);
    hipLaunchKernelGGL(k_sort_keys, dim3(1), dim3(SORT_THREADS), 0, st, keys, counts, ni, L.cap_pad, (int2*)nullptr, 0,
                       (int*)nullptr, 0);
    hipLaunchKernelGGL(k_nms_plain_gather, dim3((ni + 255) / 256), dim3(256), 0, st, boxes, keys, ni, det);
    hipLaunchKernelGGL(k_nms_greedy<16>, dim3(1, 1), dim3(1024), 0, st, det, keys2, L.m_pad, L.m_cap, m_arr, range,
                       (const int2*)nullptr, 1, 0.0f, round_down_to_float(iou_thres), ni, keep, spill);
    hipLaunchKernelGGL(k_nms_collect, dim3(1), dim3(1024), 0, st, det, keep, keys, L.cap_pad, L.m_cap, counts, m_arr,
                       (const int*)nullptr, ni, (float*)nullptr, (long long*)keep_index, keep_count, (int*)nullptr,
                       (int2*)nullptr, (int*)nullptr, (int*)nullptr, ni);
    return dh_launch_status();
}

}  // namespace

extern "C" int dh_nms_plain(const float* boxes, const float* scores, int64_t n, double iou_thres, int64_t* keep_index,
                            int* keep_count, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || n >= (1ll << 31) || !keep_count) return DH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return dh_fill_async(keep_count, 0, sizeof(int), st);
    if (!boxes || !scores || !keep_index || !workspace) return DH_ERR_ARG;
    return nms_plain_run(boxes, scores, (int)n, iou_thres, keep_index, keep_count, workspace, workspace_bytes, st);
}

// ------------------------------------------------------------------------------------------ nms() without any allocation
// The same launches as dh_nms_plain on a workspace that is part of the library's own image: a file-scope device array
// (one copy per GPU, at its own address, held until the process ends).  In host code the array's name is only a handle
// for the symbol; its device address comes from hipGetSymbolAddress, once per device, and everything below uses that.
#define NOALLOC_MAX_BOXES 32768
#define NOALLOC_SCRATCH_BYTES (2304u * 1024u)
#define NOALLOC_MAX_DEVICES 64

static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

namespace {
std::mutex g_scratch_lock;
void* g_scratch_ptr[NOALLOC_MAX_DEVICES];      // device address of `scratch` on each GPU (null: not looked up yet)
}  // namespace

extern "C" int dh_nms_plain_noalloc(const float* boxes, const float* scores, int64_t n, double iou_thres, int64_t* keep_index,
                                    int* keep_count, void* stream)
{
    if (n < 0 || !keep_count) return DH_ERR_ARG;
    if (n > NOALLOC_MAX_BOXES) return DH_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return dh_fill_async(keep_count, 0, sizeof(int), st);
    if (!boxes || !scores || !keep_index) return DH_ERR_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) return DH_ERR_ARG;
    // Every GPU holds its own copy of the array, and the launches below go to the current device: a stream that lives on
    // another GPU would pair that GPU's kernels with this GPU's array.  Refused instead of guessed at.
    if (st) {
        hipDevice_t of_stream = -1;
        if (hipStreamGetDevice(st, &of_stream) == hipSuccess && (int)of_stream != dev) return DH_ERR_ARG;
    }
    void* ws = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_scratch_lock);
        if (!g_scratch_ptr[dev]) {          // first call on this device (which is current): look the array's address up
            void* found = nullptr;
            if (hipGetSymbolAddress(&found, HIP_SYMBOL(scratch)) != hipSuccess || !found) return DH_ERR_LAUNCH;
            g_scratch_ptr[dev] = found;
        }
        ws = g_scratch_ptr[dev];
    }
    if (make_layout(1, (int)n, (int)n, 1, false).total > NOALLOC_SCRATCH_BYTES) return DH_ERR_TOO_LARGE;
    return nms_plain_run(boxes, scores, (int)n, iou_thres, keep_index, keep_count, ws, NOALLOC_SCRATCH_BYTES, st);
}
