"""ctypes binding of libdethead.so (the C ABI declared in include/dethead.h).

There is no CPU or eager-PyTorch fallback anywhere in this package: if the shared
library is missing, or a tensor is not on a ROCm device, the call raises.
"""
import ctypes
import os

import torch  # imported first on purpose: libdethead must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdethead.so")

c_void_p, c_int, c_i64, c_double, c_float, c_size_t = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_float, ctypes.c_size_t)

ERRORS = {-1: "invalid argument (shape / null pointer / unsupported size)", -2: "HIP kernel launch failed",
          -3: "pointer not 16-byte aligned", -4: "workspace too small",
          -5: "input exceeds the fixed capacity of an entry point that allocates nothing"}

# name -> (restype, argtypes); must list every symbol include/dethead.h declares
SIGNATURES = {
    "dh_bbox_iou_fwd": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, c_int,
                                c_double, c_void_p]),
    "dh_bbox_iou_bwd": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64,
                                c_i64, c_int, c_int, c_double, c_void_p]),
    "dh_centernet_box_loss_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dh_centernet_box_loss_fwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                                          c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    "dh_centernet_box_loss_bwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                                          c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                          c_void_p, c_i64, c_void_p, c_i64, c_void_p]),
    "dh_centernet_box_loss_fwd_bwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                                              c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                              c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_yolo_decode_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_yolo_decode_half": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    "dh_yolo_head_maps": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_yolo_head_maps_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dh_yolo_head_maps_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_yolo_head_maps_half": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "dh_yolo_head_maps_bwd_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dh_nms_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dh_nms_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int, c_int, c_void_p, c_int, c_int,
                               c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_nms_prepare_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_nms_batched_prepared": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int, c_int, c_void_p, c_int, c_int,
                                        c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "dh_nms_batched_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int, c_int, c_void_p, c_int, c_int,
                                   c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_nms_merge": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_double, c_int, c_void_p,
                             c_size_t, c_void_p]),
    "dh_yolo_decode_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                   c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_decode_nms_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                       c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_decode_nms_half": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                        c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_conv_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                 c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_conv_nms_half": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                      c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_conv_nms_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                     c_double, c_int, c_int, c_void_p, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_nms_plain_workspace": (c_size_t, [c_i64]),
    "dh_nms_plain": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_nms_plain_noalloc": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_void_p, c_void_p, c_void_p]),
    "dh_box_convert": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "dh_scale_clip_coords": (c_int, [c_void_p, c_i64, c_i64, c_int, c_double, c_double, c_double, c_double, c_double,
                                     c_void_p]),
    "dh_descale_pred": (c_int, [c_void_p, c_i64, c_i64, c_double, c_int, c_double, c_double, c_void_p]),
    "dh_build_targets_workspace": (c_size_t, [c_int, c_int, c_i64]),
    "dh_build_targets": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_double, c_i64, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_workspace": (c_size_t, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_i64]),
    "dh_yolo_loss_sums_offset": (c_size_t, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_i64]),
    "dh_yolo_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_fwd_bwd": (c_int, [c_void_p, c_i64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_fwd_conv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_bwd_conv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_fwd_bwd_conv": (c_int, [c_void_p, c_i64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_fwd_conv_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_bwd_conv_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_fwd_bwd_conv_f16": (c_int, [c_void_p, c_i64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_positives": (c_int, [c_void_p, c_i64, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_size_t,
                                       c_void_p, c_size_t, c_void_p]),
    "dh_yolo_loss_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "dh_yolo_loss_bwd_scaled": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_stream_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_box_iou": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p]),
    "dh_wh_iou": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p]),
    "dh_centernet_pool_nms": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "dh_centernet_decode_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dh_centernet_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_eval_match_workspace": (c_size_t, [c_i64, c_i64]),
    "dh_eval_match": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_int, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "dh_confusion_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_int, c_double, c_double,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_focal_loss_workspace": (c_size_t, [c_i64]),
    "dh_focal_loss_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_double, c_double, c_int, c_i64,
                                  c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_focal_loss_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_double, c_double, c_int, c_i64,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_regression_loss_workspace": (c_size_t, [c_i64]),
    "dh_regression_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_regression_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_anchor_metric": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_ap_classes": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dh_ap_per_class_workspace": (c_size_t, [c_i64, c_int]),
    "dh_ap_per_class": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_int, c_int, c_double,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dh_group_nms_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dh_group_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p]),
    "dh_centernet_targets": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def lib():
    """Load libdethead.so once; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C object_detection_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(rc, rc)}")


_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def raw_stream():
    """Handle (an integer) of torch's current HIP stream on the current device."""
    return _raw_stream(_get_device())


def stream():
    """torch's current HIP stream on the current device, as the void* the C ABI takes."""
    return c_void_p(_raw_stream(_get_device()))


class device_guard:
    """`with device_guard(t.device):` makes that device current for the launches inside; when it already is
    (the usual case: one process per GPU) entering and leaving cost two integer compares."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else _get_device()
        self.prev = -1

    def __enter__(self):
        prev = _get_device()
        if prev != self.idx:
            _set_device(self.idx)
            self.prev = prev
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            _set_device(self.prev)
            self.prev = -1
        return False


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def dev_f32(t, name="tensor"):
    """fp32 view of a device tensor; anything not on a ROCm device is an error (no CPU path)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: object_detection_amd only runs on ROCm devices (no CPU fallback)")
    return t if t.dtype == torch.float32 else t.float()


def head_maps(levels, name="x"):
    """The raw head maps as the kernels take them: (contiguous tensors, is_half).  A head that runs in half precision
    (every map fp16) is read as it is; anything else goes through dev_f32."""
    if levels and all(isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float16 for x in levels):
        return [x.contiguous() for x in levels], True
    return [dev_f32(x, f"{name}[{i}]").contiguous() for i, x in enumerate(levels)], False


def workspace(nbytes, device):
    """256-byte aligned scratch from torch's caching allocator."""
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class YoloHyp(ctypes.Structure):
    """Mirror of dh_yolo_hyp in include/dethead.h."""
    _fields_ = [(n, c_float) for n in ("box", "obj", "cls", "cls_pw", "obj_pw", "fl_gamma", "cp", "cn", "gr")] + [
        ("balance", c_float * 5), ("sort_obj_iou", c_int)]


_ARRAYS = {}


def int_array(values):
    """Read-only C int array of `values` (cached: the same few shapes recur every step)."""
    key = ("i",) + tuple(int(v) for v in values)
    hit = _ARRAYS.get(key)
    if hit is None:
        if len(_ARRAYS) > 256:
            _ARRAYS.clear()
        hit = _ARRAYS[key] = (c_int * (len(key) - 1))(*key[1:])
    return hit


def float_array(values):
    return (c_float * len(values))(*[float(v) for v in values])
