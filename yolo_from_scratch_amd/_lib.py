"""ctypes binding of libyfs.so (include/yfs.h).  No fallback: if the HIP library is missing the
import of any op raises, loudly."""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: libyfs.so binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# YFS_LIB_PATH: load another build of the same ABI (A/B measurements of two builds on one GPU box)
LIB_PATH = os.environ.get("YFS_LIB_PATH") or os.path.join(_HERE, "libyfs.so")
_lib = None


class YfsError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YfsError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name in ("yfs_iou_xywh_workspace_bytes", "yfs_yolov1_loss_workspace_bytes", "yfs_mse_workspace_bytes",
                     "yfs_conv2d_wgrad_workspace_bytes", "yfs_sumsq_workspace_bytes",
                     "yfs_conv2d_fwd_workspace_bytes", "yfs_conv2d_dgrad_workspace_bytes",
                     "yfs_conv2d_stem_wgrad_workspace_bytes", "yfs_single_box_loss_workspace_bytes",
                     "yfs_mask_to_boxes_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_size_t
    return _lib


KNOBS = {"igemm_tile": 0, "wgrad_tile": 1, "wgrad_splits": 2, "nms_fused": 3, "stem_kernel": 4, "igemm_persist": 5}     # include/yfs.h: yfs_knob_id


def set_knob(name, value):
    """Measurement aid (include/yfs.h, tuning knobs): force a kernel variant; 0 restores the library's own choice."""
    check(lib().yfs_set_knob(ctypes.c_int(KNOBS[name]), ctypes.c_int(int(value))), "yfs_set_knob")


def get_knob(name):
    return lib().yfs_get_knob(ctypes.c_int(KNOBS[name]))


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().yfs_last_error(buf, ctypes.c_size_t(512))
    return buf.value.decode("utf-8", "replace")


def last_kernel():
    """Kernel variant the calling thread's last conv entry point launched (include/yfs.h: yfs_last_kernel)."""
    buf = ctypes.create_string_buffer(128)
    lib().yfs_last_kernel(buf, ctypes.c_size_t(128))
    return buf.value.decode("utf-8", "replace")


def check(status, what):
    if status != 0:
        raise YfsError(f"{what} failed with status {status}: {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise YfsError("yolo_from_scratch_amd ops run on the MI355X only: got a CPU tensor "
                           "(there is no CPU fallback; the CPU oracle lives in oracle/ and is test-only)")


i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t
