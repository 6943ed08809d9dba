#!/usr/bin/env python3
"""Do the hipMemsetAsync calls of libdethead, captured by torch.cuda.graph, take effect on every replay?
Only entry points that enqueue nothing but memsets are used (nms() of zero boxes, the assignment of zero targets), so
no kernel of the library runs and nothing can address outside its buffers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _lib as L  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lib = L.lib()
    count = torch.full((64,), 7, dtype=torch.int32, device=dev)
    counts = torch.full((64,), 9, dtype=torch.int32, device=dev)
    fill = torch.zeros(4096, dtype=torch.int32, device=dev)
    anchors = torch.ones(3, 3, 2, device=dev)
    ny = L.int_array([8, 4, 2])

    def fn():
        L.check(lib.dh_nms_plain(None, None, 0, 0.5, None, L.ptr(count), None, 0, L.stream()), "dh_nms_plain")
        L.check(lib.dh_build_targets(None, 0, L.ptr(anchors), 3, 3, ny, ny, 4.0, 0, None, None, None, None, L.ptr(counts), None, 0,
                                     L.stream()), "dh_build_targets")

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    print("eager:", count[:2].tolist(), counts[:4].tolist(), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for r in range(3):
        count.fill_(7)
        counts.fill_(9)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ok = count[0].item() == 0 and counts[:3].tolist() == [0, 0, 0] and counts[3].item() == 9
        print(f"replay {r}: keep_count {count[0].item()} (7 before), level counts {counts[:4].tolist()} (9 before): "
              f"{'memsets took effect' if ok else 'MEMSETS DID NOT TAKE EFFECT'}", flush=True)
    del g, fill


if __name__ == "__main__":
    main()
