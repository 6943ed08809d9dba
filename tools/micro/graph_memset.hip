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

// Several memset nodes of different sizes in ONE graph, the buffers dirtied by a kernel between the replays: what the
// library's NMS and assignment entry points looked like when they cleared their bookkeeping with hipMemsetAsync.
__global__ void k_dirty(int* a, int na, int* b, int nb)
{
    for (int i = threadIdx.x; i < na; i += blockDim.x) a[i] = 0x5A5A5A5A;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = 0x5A5A5A5A;
}

__global__ void k_read2(const int* a, int na, const int* b, int nb, int* seen)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { seen[0] = a[0]; seen[1] = a[na - 1]; seen[2] = b[0]; seen[3] = b[nb - 1]; }
}

static int run_pair(size_t bytes_a, size_t bytes_b, hipStream_t st, int replays, int graphs)
{
    const int na = (int)(bytes_a / 4), nb = (int)(bytes_b / 4);
    int bad = 0;
    int *a[4], *b[4], *seen[4];
    hipGraphExec_t exec[4];
    for (int g = 0; g < graphs; ++g) {
        CK(hipMalloc(&a[g], bytes_a)); CK(hipMalloc(&b[g], bytes_b)); CK(hipMalloc(&seen[g], 64));
        hipGraph_t graph;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(a[g], 0, bytes_a, st));
        CK(hipMemsetAsync(b[g], 0, bytes_b, st));
        hipLaunchKernelGGL(k_read2, dim3(1), dim3(64), 0, st, a[g], na, b[g], nb, seen[g]);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec[g], graph, nullptr, nullptr, 0));
        CK(hipGraphDestroy(graph));
    }
    for (int r = 0; r < replays; ++r) {
        for (int g = 0; g < graphs; ++g) {
            hipLaunchKernelGGL(k_dirty, dim3(1), dim3(256), 0, st, a[g], na, b[g], nb);
            CK(hipGraphLaunch(exec[g], st));
        }
        CK(hipStreamSynchronize(st));
        for (int g = 0; g < graphs; ++g) {
            int host[4];
            CK(hipMemcpy(host, seen[g], sizeof(host), hipMemcpyDeviceToHost));
            const bool ok = !host[0] && !host[1] && !host[2] && !host[3];
            bad += !ok;
            printf("  two memset nodes (%zu B, %zu B), graph %d of %d, replay %d: 0x%08X 0x%08X | 0x%08X 0x%08X  %s\n", bytes_a, bytes_b,
                   g, graphs, r, host[0], host[1], host[2], host[3], ok ? "both took effect" : "A MEMSET NODE WROTE SOMETHING ELSE");
        }
    }
    for (int g = 0; g < graphs; ++g) { CK(hipGraphExecDestroy(exec[g])); CK(hipFree(a[g])); CK(hipFree(b[g])); CK(hipFree(seen[g])); }
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
    bad += run_pair(4, 12, st, 4, 1);          // a 4-byte fill followed by a 12-byte one, as the assignment entry made them
    bad += run_pair(4, 12, st, 4, 3);          // three such graphs replayed in turn
    bad += run_pair(12, 215296, st, 4, 3);
    printf("%s\n", bad ? "RESULT: captured memsets are NOT reliable" : "RESULT: captured memsets replay correctly");
    return 0;
}
