"""CPU oracle for the detection-head hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``object_detection_amd`` may import this
package; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker / CPU timing baseline.

Every function restates, in plain PyTorch-CPU / numpy / C, the arithmetic of one
function of the reference (cited file:line in each docstring) with the same
operation order, so fp32 results agree bit-for-bit with the reference on CPU.
The restatement is pinned by the fixtures under ``tests/golden/`` which were
produced by running the reference's own code (``tests/golden/make_golden.py``).
"""
