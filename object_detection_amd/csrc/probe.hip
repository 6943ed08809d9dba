// Measurement aid: a plain streaming copy, one 16-byte non-temporal load and store per thread, as many workgroups as it
// takes -- the launch shape that reaches the highest copy rate on this chip (tools/micro/copy_bw.hip: 6.1-6.7 TB/s against
// 4.3-4.5 TB/s with eight vectors per thread).  bench.py times it on the bytes one decode launch moves and reports the
// rate beside the nominal 8 TB/s, as the ceiling the memory system shows a kernel that does nothing but move data.
// No counterpart in the reference.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {
__global__ void __launch_bounds__(256) k_stream_copy(const dh_f4* __restrict__ in, dh_f4* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
}  // namespace

extern "C" int dh_stream_copy(const void* src, void* dst, size_t bytes, void* stream)
{
    if (bytes == 0) return DH_OK;
    if (!src || !dst || bytes % 16) return DH_ERR_ARG;
    if (!dh_aligned16(src) || !dh_aligned16(dst)) return DH_ERR_ALIGN;
    const size_t n = bytes / 16;
    if ((n + 255) / 256 >= (1ull << 31)) return DH_ERR_ARG;
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const dh_f4*)src, (dh_f4*)dst, n);
    return dh_launch_status();
}
