// Shared device/host helpers for the detection-head kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DH_OK 0
#define DH_ERR_ARG (-1)      // bad shape / null pointer / unsupported size
#define DH_ERR_LAUNCH (-2)   // hipGetLastError() after a launch was not hipSuccess
#define DH_ERR_ALIGN (-3)    // pointer not aligned as the kernel requires
#define DH_ERR_WORKSPACE (-4)  // workspace too small
#define DH_ERR_TOO_LARGE (-5)  // problem exceeds the fixed capacity of an entry point that allocates nothing

#define DH_WAVE 64

typedef float dh_f4 __attribute__((ext_vector_type(4)));   // native 16-byte vector (usable with the nontemporal builtins)

static inline int dh_launch_status() { return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH; }
static inline bool dh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Compute units of the current device (256 on MI355X); asked once per device, 0 if the runtime will not say.
static inline int dh_cu_count()
{
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
        cached[dev] = n;
    }
    return cached[dev];
}

// Every fill of the library goes through this kernel, not through hipMemsetAsync.  A memset captured into a HIP graph is a
// memset node; a fill kernel is a kernel node and replays like any other launch.  The NMS and loss bookkeeping start from an
// all-zero (or all-ones) state and derive indices from it, so the library does not depend on how a runtime replays memset
// nodes.  (A three-graph replay of a full-size step faulted while the fills were memsets and has replayed correctly since
// they are kernels and the consumers of the counters clamp what they read; stand-alone probes of captured memsets --
// tools/micro/graph_memset.hip, tools/graph_memset_probe.py -- replay correctly on ROCm 7.2, so the memset nodes are not
// shown to have been the cause.)
template <typename V>
__global__ void __launch_bounds__(256) dh_k_fill(V* __restrict__ p, V value, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = value;
}

static inline int dh_fill_async(void* p, int byte_value, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return DH_OK;
    if (!p) return DH_ERR_ARG;
    const uint32_t b = (uint32_t)byte_value & 0xFFu, w = b * 0x01010101u;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (((a | bytes) & 15u) == 0 && bytes / 16 / 256 < (1ull << 31)) {
        const size_t n = bytes / 16;
        hipLaunchKernelGGL(dh_k_fill<uint4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint4*)p, make_uint4(w, w, w, w), n);
    } else if (((a | bytes) & 3u) == 0 && bytes / 4 / 256 < (1ull << 31)) {
        const size_t n = bytes / 4;
        hipLaunchKernelGGL(dh_k_fill<uint32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, w, n);
    } else if (bytes / 256 < (1ull << 31)) {
        hipLaunchKernelGGL(dh_k_fill<unsigned char>, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, st, (unsigned char*)p,
                           (unsigned char)b, bytes);
    } else {
        return DH_ERR_ARG;
    }
    return dh_launch_status();
}

// Exact unsigned division by a runtime constant for numerators < 2^31:
// q = (n * magic) >> 32 >> shift  (round-up method).  Built on the host, used on the device.
struct dh_fastdiv {
    uint32_t magic;
    uint32_t shift;
    uint32_t d;
};

static inline dh_fastdiv dh_make_fastdiv(uint32_t d)
{
    dh_fastdiv f;
    f.d = d;
    if (d == 1) { f.magic = 0; f.shift = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                 // l = ceil(log2 d)
    const uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.magic = (uint32_t)m;
    f.shift = l;
    return f;
}

// floor(n / d) for n < 2^32 (Granlund-Montgomery: t = mulhi(n,m); q = (t + ((n - t) >> 1)) >> (l-1)).
__device__ __forceinline__ uint32_t dh_div(uint32_t n, const dh_fastdiv f)
{
    if (f.d == 1) return n;
    const uint32_t t = __umulhi(n, f.magic);
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

__device__ __forceinline__ float dh_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double dh_wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 1/(1+exp(-x)) with the hardware exp2/rcp units (relative error < 5e-7).
__device__ __forceinline__ float dh_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
