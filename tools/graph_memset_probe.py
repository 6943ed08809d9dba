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


def raw_memset_in_graphs():
    """hipMemsetAsync itself (called through ctypes on the HIP runtime torch has loaded), captured next to torch kernels:
    one graph with a single 215 296-byte fill, replayed in turn with two graphs that hold no memset, then a graph with
    a 4-byte and a 12-byte fill.  Buffers are set to 0x5A before every replay and read back after it."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    dev = torch.device("cuda", 0)
    a, b, c = (torch.zeros(1 << 16, device=dev) for _ in range(3))
    buf = torch.empty(215296, dtype=torch.uint8, device=dev)
    small = torch.empty(64, dtype=torch.int32, device=dev)

    def capture(fn):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def f_one():
        assert hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel(), L.raw_stream()) == 0
        c.add_(1)

    def f_two():
        assert hip.hipMemsetAsync(small.data_ptr(), 0, 4, L.raw_stream()) == 0
        assert hip.hipMemsetAsync(small.data_ptr() + 16, 0, 12, L.raw_stream()) == 0
        c.add_(1)

    ga, gb = capture(lambda: a.mul_(1.0)), capture(lambda: b.add_(0.0))
    g1, g2 = capture(f_one), capture(f_two)
    bad = 0
    for r in range(8):
        buf.fill_(0x5A)
        small.fill_(0x5A5A5A5A)
        torch.cuda.synchronize()
        ga.replay(); gb.replay(); g1.replay(); g2.replay()
        torch.cuda.synchronize()
        vals = torch.unique(buf).tolist()
        sm = [hex(v & 0xFFFFFFFF) for v in small[:8].tolist()]
        ok = vals == [0] and small[0].item() == 0 and small[4:7].tolist() == [0, 0, 0]
        bad += not ok
        print(f"raw memsets, replay {r}: bytes of the 215 296-byte buffer {vals[:4]}, words of the small one {sm}: "
              f"{'ok' if ok else 'WRONG FILL'}", flush=True)
    print("RESULT:", "captured hipMemsetAsync is NOT reliable" if bad else "captured hipMemsetAsync replayed correctly", flush=True)
    del ga, gb, g1, g2


if __name__ == "__main__" and "--raw" in sys.argv:
    raw_memset_in_graphs()
