"""HIP-graph capture and replay of the three stages of a step (decode, assignment + loss forward/backward, NMS), each its own
graph with its own memory pool, replayed in turn and mixed with eager calls: every replay reproduces the eager results.
Every fill the library needs is a kernel node (not a captured hipMemsetAsync), and the consumers of the counters clamp what
they read, so the bookkeeping of NMS and loss starts from the right state on every replay."""
import pytest

pytestmark = pytest.mark.gpu


def test_three_stage_graphs_replay_like_eager():
    from tools.graph_replay import main
    ok = main(bs=4, replays=6)
    assert all(ok.values()), ok
    assert set(ok) == {"decode", "loss", "nms", "three graphs in turn", "graphs mixed with eager calls"}


def test_decode_and_nms_capture_with_a_device_resident_stride():
    """A model moved to the device keeps Detect.stride there (yolo.py:231-236).  After one eager call the decode and the NMS
    launch read no device tensor on the host any more: they can be captured, and the replay gives the eager results."""
    import numpy as np
    import torch
    from object_detection_amd.general import nms_launch
    from object_detection_amd.yolo import detect_decode
    from oracle import yolo as oy
    from tests._synth import synth_head
    rs = np.random.RandomState(3)
    levels = [torch.from_numpy(x).cuda() for x in synth_head(rs, 2, 128, 80)]
    anchors = oy.grid_anchors().cuda()
    stride = torch.tensor(oy.YOLOV5S_STRIDES).cuda()
    want = detect_decode(levels, anchors, stride)                       # eager: reads the stride once
    rows_e, meta_e, _ = nms_launch(want, 0.001, 0.45, max_det=300, cap=want.shape[1])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):                      # a device-to-host copy in here would raise
            got = detect_decode(levels, anchors, stride)
            rows, meta, _ = nms_launch(got, 0.001, 0.45, max_det=300, cap=got.shape[1])
        for _ in range(3):
            graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    m, me, r, re_ = meta.cpu().reshape(-1), meta_e.cpu().reshape(-1), rows.cpu(), rows_e.cpu()
    for i in range(2):          # images are packed in the order they finish: compare image by image
        (f, n), (fe, ne) = (int(m[2 * i]), int(m[2 * i + 1])), (int(me[2 * i]), int(me[2 * i + 1]))
        assert n == ne > 0 and torch.equal(r[f:f + n], re_[fe:fe + ne])
