"""HIP-graph capture and replay of the three stages of a step (decode, assignment + loss forward/backward, NMS), each its own
graph with its own memory pool, replayed in turn and mixed with eager calls: every replay reproduces the eager results.
Every fill the library needs is a kernel node (a captured hipMemsetAsync is a memset node, and those replay with wrong
fill values on this runtime), so the bookkeeping of NMS and loss starts from the right state on every replay."""
import pytest

pytestmark = pytest.mark.gpu


def test_three_stage_graphs_replay_like_eager():
    from tools.graph_replay import main
    ok = main(bs=4, replays=6)
    assert all(ok.values()), ok
    assert set(ok) == {"decode", "loss", "nms", "three graphs in turn", "graphs mixed with eager calls"}
