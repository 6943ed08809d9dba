"""dh_nms_plain_noalloc: the nms() entry point that allocates nothing (its workspace is a fixed device array inside
the library) against dh_nms_plain on a caller-owned workspace, through the C ABI; bit-exact (index work)."""
import numpy as np
import pytest
import torch

from object_detection_amd import _lib as L

pytestmark = pytest.mark.gpu


def _boxes(rs, n):
    xy = rs.uniform(0, 600, (n, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(200), (n, 2)))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = rs.permutation(n).astype(np.float32) / n          # distinct scores
    return torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()


def _plain(boxes, scores, thr):
    lib = L.lib()
    n = boxes.shape[0]
    ws_bytes = lib.dh_nms_plain_workspace(n)
    ws = L.workspace(ws_bytes, boxes.device)
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=boxes.device)
    count = torch.full((1,), -1, dtype=torch.int32, device=boxes.device)
    L.check(lib.dh_nms_plain(L.ptr(boxes), L.ptr(scores), n, thr, L.ptr(keep), L.ptr(count), L.ptr(ws), ws_bytes, L.stream()),
            "dh_nms_plain")
    return keep[:int(count.item())].clone()


def _noalloc(boxes, scores, thr):
    n = boxes.shape[0]
    keep = torch.full((max(n, 1),), -1, dtype=torch.int64, device=boxes.device)
    count = torch.full((1,), -1, dtype=torch.int32, device=boxes.device)
    L.check(L.lib().dh_nms_plain_noalloc(L.ptr(boxes), L.ptr(scores), n, thr, L.ptr(keep), L.ptr(count), L.stream()),
            "dh_nms_plain_noalloc")
    return keep[:int(count.item())].clone()


def test_noalloc_nms_matches_nms_on_two_calls_in_a_row():
    rs = np.random.RandomState(90)
    big, small = _boxes(rs, 20000), _boxes(rs, 700)          # the second call is smaller: the array keeps the first one's data
    want_big, want_small = _plain(*big, 0.5), _plain(*small, 0.45)
    torch.cuda.synchronize()
    got_big = _noalloc(*big, 0.5)
    got_small = _noalloc(*small, 0.45)
    assert got_big.numel() == want_big.numel() > 1000 and torch.equal(got_big, want_big)
    assert got_small.numel() == want_small.numel() > 50 and torch.equal(got_small, want_small)
    # and against the CPU oracle's greedy NMS (the published torchvision algorithm; no torchvision here to pin it further)
    from oracle import nms as onms
    for (b, s), thr, got in ((big, 0.5, got_big), (small, 0.45, got_small)):
        assert torch.equal(got.cpu(), onms.nms(b.cpu(), s.cpu(), thr))
    # once more the other way round, and the largest size the array holds
    assert torch.equal(_noalloc(*small, 0.45), want_small)
    top = _boxes(rs, 32768)
    assert torch.equal(_noalloc(*top, 0.6), _plain(*top, 0.6))
    assert torch.equal(_noalloc(*big, 0.5), want_big)


def test_noalloc_nms_allocates_nothing_and_rejects_what_does_not_fit():
    rs = np.random.RandomState(91)
    boxes, scores = _boxes(rs, 5000)
    _noalloc(boxes, scores, 0.5)                                # first call on the device: address look-up
    keep = torch.empty(5000, dtype=torch.int64, device="cuda")
    count = torch.empty(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    stats0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(5):
        assert L.lib().dh_nms_plain_noalloc(L.ptr(boxes), L.ptr(scores), 5000, 0.5, L.ptr(keep), L.ptr(count), L.stream()) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0                # no device memory taken from the driver
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == stats0      # nor from torch's allocator
    rc = L.lib().dh_nms_plain_noalloc(L.ptr(boxes), L.ptr(scores), 32769, 0.5, L.ptr(keep), L.ptr(count), L.stream())
    assert rc == -5
    with pytest.raises(RuntimeError, match="fixed capacity"):
        L.check(rc, "dh_nms_plain_noalloc")
    # n = 0: count 0, nothing else touched
    assert L.lib().dh_nms_plain_noalloc(None, None, 0, 0.5, None, L.ptr(count), L.stream()) == 0
    assert int(count.item()) == 0
