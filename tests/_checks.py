"""Tolerance checks shared by the GPU tests and the smoke run."""
import numpy as np
import torch


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def decoded_close(got, want, rows_per_level, strides, rtol=1e-5, atol=1e-7):
    """A decoded (bs, A, 5+nc) tensor against its reference, channel group by channel group.

    xy = (2*sigmoid - 0.5 + cell) * stride cancels near a cell's origin, so its error is bounded in pixels:
    1e-5 of the stride (the scale of the terms that cancel), plus two fp32 ulps of the result itself (at 1280 px one
    ulp of a coordinate is 1.2e-4 px, more than 1e-5 of the finest stride).  wh, objectness and class scores are products of
    positive terms: 1e-5 relative, with an absolute floor of 1e-7 (far below any threshold the scores meet).
    rows_per_level: na*ny*nx of each level, in order; strides: the levels' strides."""
    got, want = _np(got), _np(want)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    assert sum(rows_per_level) == got.shape[1], (rows_per_level, got.shape)
    first = 0
    for n, s in zip(rows_per_level, strides):
        g, w = got[:, first:first + n], want[:, first:first + n]
        np.testing.assert_allclose(g[..., 0:2], w[..., 0:2], rtol=2.4e-7, atol=1e-5 * float(s), err_msg=f"xy, stride {s}")
        np.testing.assert_allclose(g[..., 2:4], w[..., 2:4], rtol=rtol, atol=atol, err_msg=f"wh, stride {s}")
        np.testing.assert_allclose(g[..., 4:], w[..., 4:], rtol=rtol, atol=atol, err_msg=f"obj/cls, stride {s}")
        first += n


def level_rows(levels):
    """na*ny*nx of every (bs, na, ny, nx, no) map."""
    return [int(x.shape[1] * x.shape[2] * x.shape[3]) for x in levels]


GRAD_GROUPS = (("box", slice(0, 4)), ("objectness", slice(4, 5)), ("class", slice(5, None)))


def grads_close(got, want, want64=None, rtol=1e-5, floor=1e-6, suspects=0.01, atol=0.0, label=""):
    """dL/dp of ONE level, (bs, na, ny, nx, 5+nc), against the reference gradient, channel group by channel group.

    The three groups differ in magnitude by orders (the dense objectness gradient is a thousand times smaller than a box
    gradient), so a tolerance taken from the whole tensor says nothing about the small ones.  Each group (box 0-3,
    objectness 4, class 5+) is compared on its own:  |got - want| <= rtol*|want| + floor * (median |want| of that
    group's non-zero entries on this level).  Entries where want is exactly zero (every channel but objectness off the
    assigned cells) must be exactly zero.

    An entry that fails that test is one where fp32 rounding shows: a difference of nearly equal terms (sigmoid(x) - tobj,
    the sum of the CIoU terms), or the focal loss, whose formula loses 1e-4 to rounding in the reference itself.  Such an
    entry must be as close to the float64 evaluation `want64` of the same formulas as the fp32 reference is:
    |got - w64| <= rtol*|w64| + 4*|want - w64|.  Only on assigned cells (cells whose box gradient is non-zero) may an
    entry miss that too, by at most floor * (largest |w64| among the assigned cells' entries of the group: the scale of
    the terms that cancel), and at most `suspects` of those entries may.  want64 may be a callable
    (evaluated only when needed) or None (no such entries allowed).  atol: an absolute term on top, zero for fp32
    gradients; the fp16-gradient tests set it to the spacing of fp16 subnormals, which is what bounds tiny elements there."""
    got, want = _np(got), _np(want)
    assert got.shape == want.shape and got.ndim == 5, (label, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{label}: non-finite gradient"
    assigned = (want[..., 0:4] != 0).any(-1)
    w64 = None
    for name, sl in GRAD_GROUPS:
        g, w = got[..., sl].astype(np.float64), want[..., sl].astype(np.float64)
        if w.size == 0:
            continue
        nz = w != 0
        stray = (~nz) & (g != 0) & ~assigned[..., None]
        assert not stray.any(), f"{label} {name}: {int(stray.sum())} entries off the assigned cells should be exactly zero"
        scale = float(np.median(np.abs(w[nz]))) if nz.any() else 0.0
        bad = np.abs(g - w) > rtol * np.abs(w) + floor * scale + atol
        if not bad.any():
            continue
        if w64 is None:
            assert want64 is not None, f"{label} {name}: {int(bad.sum())} entries outside rtol {rtol} and no float64 evaluation given"
            w64 = _np(want64() if callable(want64) else want64)
            assert w64.shape == want.shape and w64.dtype == np.float64
        t = w64[..., sl]
        err, allowed = np.abs(g - t), rtol * np.abs(t) + 4 * np.abs(w - t) + atol
        still = bad & (err > allowed)                       # not even as close to the float64 value as the reference is
        if not still.any():
            continue
        on = np.broadcast_to(assigned[..., None], w.shape)
        worst = float((err / np.maximum(np.abs(t), 1e-300))[still & ~on].max()) if (still & ~on).any() else 0.0
        assert not (still & ~on).any(), (f"{label} {name}: {int((still & ~on).sum())} entries off the assigned cells are further from the "
                                         f"float64 value than the fp32 reference allows (worst {worst:.3g} relative)")
        cancel_scale = float(np.abs(t[on]).max())
        ok = err[still] <= allowed[still] + floor * cancel_scale
        assert ok.all(), (f"{label} {name}: {int((~ok).sum())} of {int(still.sum())} ill-conditioned entries are further from the float64 "
                          f"value than the fp32 reference allows")
        assert still.sum() <= max(2, suspects * on.sum()), \
            f"{label} {name}: {int(still.sum())} of {int(on.sum())} assigned entries needed the bound of the cancelling terms"


def all_grads_close(got_levels, want_levels, want64_levels=None, **kw):
    """grads_close over the levels of a head; want64_levels: list, or a callable returning the list (evaluated once, lazily)."""
    cache = {}

    def lazy(l):
        def get():
            if "v" not in cache:
                cache["v"] = want64_levels() if callable(want64_levels) else want64_levels
            return cache["v"][l]
        return get
    assert len(got_levels) == len(want_levels)
    for l, (g, w) in enumerate(zip(got_levels, want_levels)):
        grads_close(g, w, None if want64_levels is None else lazy(l), label=f"level {l}", **kw)


def loss_grads64(oracle, levels, targets, upstream=1.0):
    """The float64 evaluation of a YoloLossOracle's gradients on fp32 inputs (numpy or tensors), as a lazy callable.
    The oracle object is copied so that autobalance state is not advanced."""
    import copy

    def run():
        o = copy.deepcopy(oracle)
        p = [torch.as_tensor(_np(x)).double().requires_grad_(True) for x in levels]
        loss, _ = o(p, torch.as_tensor(_np(targets)))
        return [g.numpy() for g in torch.autograd.grad(loss * upstream, p)]
    return run


def conv_grad_as_maps(g, na):
    """A gradient in convolution layout (bs, na*no, ny, nx) as the (bs, na, ny, nx, no) array the map layout holds."""
    g = _np(g)
    bs, ch, ny, nx = g.shape
    return np.ascontiguousarray(g.reshape(bs, na, ch // na, ny, nx).transpose(0, 1, 3, 4, 2))
