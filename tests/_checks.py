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
