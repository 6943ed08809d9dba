"""MI355X-native detection-head hot path: box decode, IoU family, target assignment, batched NMS.

Python entry points keep the names and signatures of the reference's callables
(vkosolapov/object_detection); each is a thin wrapper over a C-ABI entry of libdethead.so.
"""
from ._lib import LIB_PATH, lib  # noqa: F401

__all__ = ["lib", "LIB_PATH"]
