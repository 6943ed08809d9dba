"""Host-side logic of the Python layer that needs no GPU: how the packed NMS rows become per-image views, and how the
loss tells lazy head maps (permuted views of the convolution outputs) from ordinary tensors."""
import torch


def test_nms_views_packed_and_with_gaps():
    from object_detection_amd.general import nms_views
    rows = torch.arange(10 * 6, dtype=torch.float32).view(10, 6)
    bs = 3
    # images complete in the order 2, 0, 1: image 2 owns rows 0-2, image 0 rows 3-6, image 1 none
    host = [3, 4, 0, 0, 0, 3] + [7, 0, 5] + [0, 0, 0]
    views, cap = nms_views(rows, host, 100)
    assert cap == 100
    assert torch.equal(views[2], rows[0:3]) and torch.equal(views[0], rows[3:7]) and views[1].shape == (0, 6)
    # merge-NMS dropped rows: image 2 keeps 2 of its 3 rows, image 0 keeps 1 of 4 -- each still starts at its first row
    host = [3, 1, 0, 0, 0, 2] + [7, 0, 5] + [0, 0, 0]
    views, _ = nms_views(rows, host, 100)
    assert torch.equal(views[2], rows[0:2]) and torch.equal(views[0], rows[3:4]) and views[1].shape == (0, 6)
    # more candidates than capacity -> ask for that capacity; a needs-full flag -> ask for the next attempt
    assert nms_views(rows, [0, 0, 0, 0, 0, 0] + [7, 500, 5] + [0, 0, 0], 100) == (None, 500)
    assert nms_views(rows, [0, 0, 0, 0, 0, 0] + [7, 0, 5] + [0, 1, 0], 100) == (None, -1)


def test_lazy_maps_are_recognised_by_layout_only_on_the_device():
    from object_detection_amd.yolo_loss import conv_layout_of
    conv = torch.randn(2, 3 * 7, 4, 5)
    view = conv.view(2, 3, 7, 4, 5).permute(0, 1, 3, 4, 2)
    assert view.shape == (2, 3, 4, 5, 7) and not view.is_contiguous()
    assert conv_layout_of([view]) is None                       # a CPU tensor never takes the kernel path
    assert conv_layout_of([view.contiguous()]) is None
    assert conv_layout_of([]) is None
