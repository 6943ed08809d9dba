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


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` starts N ranks itself; on a machine with fewer devices it says so and stops (it must
    never report n_gpus 1 for a run that asked for more)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    n = max(torch.cuda.device_count() + 1, 2)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert f"needs {n} devices" in r.stderr
    assert r.stdout.strip() == ""
    # a launcher whose WORLD_SIZE disagrees with --gpus is an error, too
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # the harness of the 1280 px inference sweep launches its ranks the same way
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_sweep.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and f"bench_sweep.py --gpus {n} needs {n} devices" in r.stderr and r.stdout.strip() == ""


def test_nms_attempt_order():
    """small -> one workgroup per image on the selected prefix -> (separate launches on that prefix) -> full sort."""
    from object_detection_amd import general as G

    class Rows:
        pass
    r = Rows()
    r._dh_attempt = ("k", G.K_SMALL)
    assert G._next_attempt(r, -1) == G.K_CHAIN
    r._dh_attempt = ("k", G.K_CHAIN)
    assert G._next_attempt(r, -2) == G.K_SELECT and G._next_attempt(r, -1) == 0
    r._dh_attempt = ("k", G.K_SELECT)
    assert G._next_attempt(r, -1) == 0 and G._next_attempt(r, -2) == 0
    # flags: 1 = the prefix did not yield max_det rows (full sort next), 2 = image left to the separate launches
    bs = 3
    host = [0, 0] * bs + [10, 10, 10] + [0, 2, 0]
    assert G.nms_views(r, host, 100) == (None, -2)
    host = [0, 0] * bs + [10, 10, 10] + [1, 2, 0]
    assert G.nms_views(r, host, 100) == (None, -1)
    host = [0, 0] * bs + [10, 500, 10] + [0, 0, 0]
    assert G.nms_views(r, host, 100) == (None, 500)
