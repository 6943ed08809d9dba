"""GPU parity of the Detect head after its convolutions: the layout change of yolo.py:54, fused with the decode of
yolo.py:56-70 in eval mode, and its backward.  The layout change moves floats without arithmetic, so everything about
it is compared bit for bit: against the oracle's restatement (view / permute / contiguous on the CPU), against torch's
own permute on the device, and — for the decoded tensor and the two compact objectness planes — against the
two-step path (maps, then detect_decode), whose parity with the oracle other tests establish."""
import numpy as np
import pytest
import torch

from oracle import yolo as oy
from tests._checks import decoded_close, level_rows

pytestmark = pytest.mark.gpu


def _convs(seed, bs, na, no, shapes, device="cuda"):
    rs = np.random.RandomState(seed)
    return [torch.from_numpy(rs.normal(0, 2, (bs, na * no, ny, nx)).astype(np.float32)).to(device) for ny, nx in shapes]


CASES = [
    # bs, na, no, [(ny, nx)]
    (2, 3, 85, [(8, 8), (4, 4), (2, 2)]),            # tiles far smaller than 64 positions
    (1, 3, 85, [(80, 80), (40, 40), (20, 20)]),      # 640 px: 400 = 6 * 64 + 16 leaves a ragged last tile
    (3, 3, 85, [(48, 80), (24, 40), (12, 20)]),      # rectangular inference shape
    (2, 3, 12, [(12, 12), (6, 6), (3, 3)]),          # even no: LDS writes with bank conflicts, same result
    (2, 3, 7, [(5, 7), (3, 3)]),                     # ny*nx*no not a multiple of 4 -> scalar path, two levels
    (1, 2, 6, [(9, 9)]),                             # no < 7 -> scalar path, one level, two anchors
    (2, 1, 256, [(10, 10)]),                         # largest channel count a tile holds
]


@pytest.mark.parametrize("bs,na,no,shapes", CASES)
def test_head_maps_equal_oracle_and_torch_permute(bs, na, no, shapes):
    from object_detection_amd.yolo import head_maps
    convs = _convs(1, bs, na, no, shapes)
    got = head_maps(convs, na)
    want = oy.head_maps([c.cpu() for c in convs], na)
    for g, w, c in zip(got, want, convs):
        assert g.shape == w.shape and g.is_contiguous()
        assert torch.equal(g.cpu(), w)
        assert torch.equal(g, c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous())


@pytest.mark.parametrize("bs,na,no,shapes", [c for c in CASES if len(c[3]) <= 3 and c[2] <= 85])
def test_head_maps_decode_equals_two_steps(bs, na, no, shapes):
    from object_detection_amd import yolo
    nl = len(shapes)
    convs = _convs(2, bs, na, no, shapes)
    anchors = (torch.arange(nl * na * 2, dtype=torch.float32).view(nl, na, 2) * 0.37 + 1.1).cuda()
    stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    dec, maps = yolo.head_maps_decode(convs, na, anchors, stride)
    ref_maps = [c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]
    ref_dec = yolo.detect_decode(ref_maps, anchors, stride)
    for m, r in zip(maps, ref_maps):
        assert torch.equal(m, r)
    assert torch.equal(dec, ref_dec)
    # the compact planes registered for the fused call are the ones the two-step call registers
    assert torch.equal(yolo.objectness_plane(dec), yolo.objectness_plane(ref_dec))
    assert torch.equal(yolo.objectness_logits(maps), yolo.objectness_logits(ref_maps))
    assert torch.equal(yolo.objectness_plane(dec), dec[:, :, 4])
    want = oy.detect_decode(oy.head_maps([c.cpu() for c in convs], na), anchors.cpu(), stride.tolist())
    decoded_close(dec, want, level_rows(maps), stride.tolist())


@pytest.mark.parametrize("bs,na,no,shapes", CASES)
def test_head_maps_backward_is_the_inverse_layout_change(bs, na, no, shapes):
    from object_detection_amd.yolo import head_maps
    convs = [c.requires_grad_() for c in _convs(3, bs, na, no, shapes)]
    maps = head_maps(convs, na)
    rs = np.random.RandomState(4)
    ups = [torch.from_numpy(rs.normal(0, 1, tuple(m.shape)).astype(np.float32)).cuda() for m in maps]
    torch.autograd.backward(maps, ups)
    for c, u in zip(convs, ups):
        # autograd through view / permute / contiguous on the CPU
        ref = c.detach().cpu().requires_grad_()
        oy.head_maps([ref], na)[0].backward(u.cpu())
        assert c.grad.is_contiguous() and torch.equal(c.grad.cpu(), ref.grad)


HALF_CASES = [c for c in CASES if c[2] <= 85] + [(2, 3, 85, [(5, 5), (3, 3)])]     # the last: odd ny*nx, positions move one by one


@pytest.mark.parametrize("bs,na,no,shapes", HALF_CASES)
def test_half_precision_head(bs, na, no, shapes):
    """fp16 convolution outputs (model.half(), val.py:115,129): fp16 maps equal torch's permute bit for bit, the decoded
    tensor equals detect_decode on those fp16 maps (itself equal to the fp32 decode of the widened maps), and the
    backward is the inverse layout change on fp16 gradients."""
    from object_detection_amd import yolo
    nl = len(shapes)
    convs = [c.half() for c in _convs(6, bs, na, no, shapes)]
    anchors = (torch.arange(nl * na * 2, dtype=torch.float32).view(nl, na, 2) * 0.37 + 1.1).cuda()
    stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    dec, maps = yolo.head_maps_decode(convs, na, anchors, stride)
    ref_maps = [c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]
    for m, r in zip(maps, ref_maps):
        assert m.dtype == torch.float16 and torch.equal(m, r)
    assert dec.dtype == torch.float32
    assert torch.equal(dec, yolo.detect_decode(ref_maps, anchors, stride))
    assert torch.equal(dec, yolo.detect_decode([r.float() for r in ref_maps], anchors, stride))
    for m, w in zip(yolo.head_maps(convs, na), oy.head_maps([c.cpu() for c in convs], na)):
        assert torch.equal(m.cpu(), w)
    leaf = [c.clone().requires_grad_() for c in convs]
    out = yolo.head_maps(leaf, na)
    torch.autograd.backward(out, [o.detach() for o in out])
    for c, l in zip(convs, leaf):
        assert l.grad.dtype == torch.float16 and torch.equal(l.grad, c)


def test_device_tensors_never_take_the_torch_copy(monkeypatch):
    """On a ROCm device fp32 / fp16 convolution outputs always go through the kernels; torch's permute copy is kept only
    for what the reference also runs elsewhere (Model.__init__ sends a CPU tensor through Detect to derive the strides,
    yolo.py:107-113)."""
    from object_detection_amd import yolo

    def boom(*a, **k):
        raise AssertionError("torch permute copy used for device tensors")
    monkeypatch.setattr(yolo, "_torch_head_maps", boom)
    for dtype in (torch.float32, torch.float16):
        convs = [c.to(dtype) for c in _convs(8, 2, 3, 85, [(8, 8), (4, 4), (2, 2)])]
        yolo.head_maps(convs, 3)
        yolo.head_maps_decode(convs, 3, oy.grid_anchors().cuda(), torch.tensor(oy.YOLOV5S_STRIDES))
    monkeypatch.undo()
    cpu = [c.cpu() for c in _convs(8, 1, 3, 85, [(4, 4)])]
    assert torch.equal(yolo.head_maps(cpu, 3)[0], oy.head_maps(cpu, 3)[0])          # the training branch on the CPU
    with pytest.raises(RuntimeError):                                              # the decode has no CPU path
        yolo.head_maps_decode(cpu, 3, oy.grid_anchors()[:1], torch.tensor([8.0]))


def test_head_maps_backward_with_an_unused_level():
    from object_detection_amd.yolo import head_maps
    convs = [c.requires_grad_() for c in _convs(5, 2, 3, 85, [(8, 8), (4, 4)])]
    maps = head_maps(convs, 3)
    maps[1].sum().backward()
    assert torch.equal(convs[1].grad, torch.ones_like(convs[1]))
    assert torch.equal(convs[0].grad, torch.zeros_like(convs[0]))


def test_detect_module_eval_and_train_match_the_reference_forward():
    """Detect.forward (yolo.py:49-70) end to end: same convolution weights, reference arithmetic restated with torch
    on the device for the layout change, the oracle's decode on the CPU for the rest."""
    from object_detection_amd.yolo import Detect
    torch.manual_seed(7)
    anchors_px = oy.YOLOV5S_ANCHORS_PX
    det = Detect(nc=80, anchors=anchors_px, ch=(16, 32, 64)).cuda()
    det.stride = torch.tensor(oy.YOLOV5S_STRIDES)
    det.anchors /= det.stride.view(-1, 1, 1).cuda()
    feats = [torch.randn(2, c, 64 // s, 96 // s, device="cuda") for c, s in zip((16, 32, 64), (8, 16, 32))]
    with torch.no_grad():
        convs = [m(f) for m, f in zip(det.m, feats)]
    ref_maps = oy.head_maps([c.cpu() for c in convs], 3)
    det.eval()
    with torch.no_grad():
        z, x = det([f.clone() for f in feats])
    for a, b in zip(x, ref_maps):
        assert torch.equal(a.cpu(), b)
    want = oy.detect_decode(ref_maps, det.anchors.cpu(), oy.YOLOV5S_STRIDES)
    decoded_close(z, want, level_rows(ref_maps), oy.YOLOV5S_STRIDES)
    det.train()
    x = det([f.clone().requires_grad_() for f in feats])
    for a, b in zip(x, ref_maps):
        assert a.requires_grad and torch.equal(a.detach().cpu(), b)
    sum((v * v).sum() for v in x).backward()
    # gradient of sum(x^2) w.r.t. the convolution bias = 2 * sum over (bs, ny, nx) of the convolution output
    for m, c in zip(det.m, convs):
        torch.testing.assert_close(m.bias.grad, 2 * c.sum((0, 2, 3)), rtol=1e-4, atol=1e-4)


def test_head_maps_full_size_round_trip():
    """bs 64 @ 640 px (548 MB per direction): backward(forward(x)) == x, forward == torch's permute, and the fused decode
    equals detect_decode on those maps; all bit for bit."""
    from object_detection_amd import yolo
    g = torch.Generator(device="cuda").manual_seed(11)
    convs = [torch.randn(64, 255, n, n, device="cuda", generator=g) for n in (80, 40, 20)]
    anchors = oy.grid_anchors().cuda()
    stride = torch.tensor(oy.YOLOV5S_STRIDES)
    dec, maps = yolo.head_maps_decode(convs, 3, anchors, stride)
    for c, m in zip(convs, maps):
        assert torch.equal(m, c.view(64, 3, 85, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous())
    assert torch.equal(dec, yolo.detect_decode(maps, anchors, stride))
    leaf = [c.clone().requires_grad_() for c in convs]
    out = yolo.head_maps(leaf, 3)
    torch.autograd.backward(out, [o.detach() for o in out])
    for c, l in zip(convs, leaf):
        assert torch.equal(l.grad, c)


@pytest.mark.parametrize("tag", ["s3_nc80", "s3_nc80_cat", "s3_nc7_rect", "s3_nc1", "s4_nc7", "s5_nc3_na2"])
def test_detect_against_the_reference_class(golden, tag):
    """tests/golden/detect.npz holds Detect.forward of the reference's own class (models/yolo.py:49-81) on the CPU:
    the outputs of its convolutions, its maps and its decoded tensor.  Maps: bit-exact (no arithmetic).  Decoded tensor:
    xy within 1e-5 * stride px, wh / objectness / classes 1e-5 relative (the device's exp against the CPU's)."""
    from object_detection_amd import yolo
    g = golden("detect")
    nc, na, nl, _, bs = (int(v) for v in g[f"{tag}_meta"])
    convs = [torch.from_numpy(g[f"{tag}_conv{l}"]).cuda() for l in range(nl)]
    anchors, strides = torch.from_numpy(g[f"{tag}_anchors"]).cuda(), torch.from_numpy(g[f"{tag}_strides"])
    want_maps = oy.head_maps([c.cpu() for c in convs], na)          # == the reference's maps (test_oracle_golden)
    want_z = torch.from_numpy(g[f"{tag}_z"])
    z, maps = yolo.head_maps_decode(convs, na, anchors, strides)
    for m, w in zip(maps, want_maps):
        assert torch.equal(m.cpu(), w)
    decoded_close(z, want_z, level_rows(maps), strides.tolist())
    z2 = yolo.detect_decode(yolo.head_maps(convs, na), anchors, strides)
    assert torch.equal(z2, z)
    # the module: same weights, the reference's attributes after an eval pass
    det = yolo.Detect(nc=nc, anchors=[[1.0] * (2 * na)] * nl, ch=[int(g[f"{tag}_w{l}"].shape[1]) for l in range(nl)]).cuda().eval()
    det.stride = strides
    with torch.no_grad():
        det.anchors.copy_(anchors)
        for l in range(nl):
            det.m[l].weight.copy_(torch.from_numpy(g[f"{tag}_w{l}"]))
            det.m[l].bias.copy_(torch.from_numpy(g[f"{tag}_b{l}"]))
        feats = [torch.zeros(bs, int(g[f"{tag}_w{l}"].shape[1]), c.shape[2], c.shape[3], device="cuda") for l, c in enumerate(convs)]
        zm, xm = det(feats)
    assert zm.shape == want_z.shape and [tuple(a.shape) for a in xm] == [tuple(w.shape) for w in want_maps]
    for l in range(nl):
        assert torch.equal(det.grid[l].cpu(), torch.from_numpy(g[f"{tag}_grid{l}"]))
        assert torch.equal(det.anchor_grid[l].cpu(), torch.from_numpy(g[f"{tag}_anchor_grid{l}"]))
