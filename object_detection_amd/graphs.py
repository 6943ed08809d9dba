"""HIP-graph capture of launch-bound kernel sequences.

The hot path is a chain of short kernels whose host-side launch cost (Python + ctypes + allocator)
is comparable to their run time; capturing a stage once and replaying it removes that cost.  Every
kernel entry of libdethead is capture-safe: no allocation, no synchronisation, only kernel launches
and stream-ordered memset/copy nodes.
"""
import torch


class GraphedStage:
    """Capture `fn()` (a closure over tensors whose storage stays put) once; `replay()` re-runs it.

    The tensors `fn` returned at capture time are kept in `.outputs`; replay refreshes their contents.
    """

    def __init__(self, fn, warmup=3, pool=None):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            self.outputs = fn()

    def pool(self):
        return self.graph.pool()

    def replay(self):
        self.graph.replay()
        return self.outputs
