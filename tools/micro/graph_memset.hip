// Does a hipMemsetAsync captured into a HIP graph take effect on every replay?  No indexing by data anywhere: the kernel
// adds 1 to word 0 of the buffer and copies three fixed words out, so nothing here can address outside its buffers.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/graph_memset.hip -o tools/micro/graph_memset && tools/micro/graph_memset
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(int* buf, int words, int* seen)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        seen[0] = atomicAdd(&buf[0], 1);      // value before the increment: 0 if the memset ran
        seen[1] = buf[1];
        seen[2] = buf[words - 1];
    }
}

static int run(size_t bytes, int value, hipStream_t st, int replays)
{
    const int words = (int)(bytes / 4);
    int *buf = nullptr, *seen = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&seen, 3 * sizeof(int) * 16));
    CK(hipMemset(buf, 0x5A, bytes));
    CK(hipDeviceSynchronize());
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(buf, value, bytes, st));
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, buf, words, seen);
    CK(hipStreamEndCapture(st, &graph));
    size_t n_nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &n_nodes));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const int want = value == 0 ? 0 : (int)0xFFFFFFFF;
    int bad = 0;
    for (int r = 0; r < replays; ++r) {
        CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        int host[3];
        CK(hipMemcpy(host, seen, sizeof(host), hipMemcpyDeviceToHost));
        const bool ok = host[0] == want && host[1] == want && host[2] == want;
        bad += !ok;
        printf("  bytes %zu value 0x%02X nodes %zu replay %d: word0 before increment 0x%08X, word1 0x%08X, last 0x%08X  %s\n", bytes, value & 0xFF,
               n_nodes, r, host[0], host[1], host[2], ok ? "memset took effect" : "MEMSET DID NOT TAKE EFFECT");
    }
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    CK(hipFree(buf));
    CK(hipFree(seen));
    return bad;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int bad = 0;
    bad += run(12, 0, st, 3);                 // the assignment's level counters
    bad += run(256, 0, st, 3);
    bad += run(215296, 0, st, 3);             // the NMS bookkeeping of a small batch
    bad += run(6451200 + 256, 0xFF, st, 3);   // the loss's owner map at bs 64
    bad += run(64u << 20, 0, st, 3);
    printf("%s\n", bad ? "RESULT: captured memsets are NOT reliable" : "RESULT: captured memsets replay correctly");
    return 0;
}
