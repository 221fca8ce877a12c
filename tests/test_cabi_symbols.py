"""The C-ABI library loads on a machine without a GPU and exports every symbol include/yfs.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "yfs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yfs_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import yolo_from_scratch_amd as y
    assert os.path.exists(y.LIB_PATH), "libyfs.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(y.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/yfs.h but not exported: {missing}"


def test_version_and_error_channel():
    import yolo_from_scratch_amd as y
    lib = y.lib()
    assert lib.yfs_version() >= 200
    # argument validation happens on the host before any launch: safe without a GPU
    rc = lib.yfs_decode_head_f32(None, ctypes.c_int64(4), ctypes.c_float(1), ctypes.c_float(1), None, None)
    assert rc == -1
    buf = ctypes.create_string_buffer(256)
    lib.yfs_last_error(buf, ctypes.c_size_t(256))
    assert b"yfs_decode_head_f32" in buf.value


def test_cpu_tensor_raises_loudly():
    import torch
    import yolo_from_scratch_amd as y
    with pytest.raises(y.YfsError):
        y.iou_xywh_torch(torch.rand(4, 4), torch.rand(4, 4))
    with pytest.raises(y.YfsError):
        y.YoloV1Loss()(torch.rand(2, 7, 7, 30), torch.rand(2, 7, 7, 30))


def _err(lib):
    buf = ctypes.create_string_buffer(512)
    lib.yfs_last_error(buf, ctypes.c_size_t(512))
    return buf.value.decode()


def test_argument_validation_status_codes_and_messages():
    """Every entry point validates on the host before any launch (safe without a GPU): unsupported shapes, missing
    workspaces and out-of-range sizes come back as negative yfs_status codes with a message naming the entry point —
    the C-side counterpart of the Python exceptions the reference's call sites raise."""
    import yolo_from_scratch_amd as y
    lib = y.lib()
    one = ctypes.c_void_p(16)                      # any non-null address: rejected before it is touched
    i64, ci, cf = ctypes.c_int64, ctypes.c_int, ctypes.c_float
    # Cin = 48 is not a multiple of 64
    rc = lib.yfs_conv2d_fwd_bf16(one, one, one, one, i64(2), ci(8), ci(8), ci(48), ci(64), ci(3), ci(1), cf(0.1), None)
    assert rc < 0 and "Cin=48" in _err(lib)
    # kernel size 4 is not in {1,3,5,7}
    rc = lib.yfs_conv2d_dgrad_bf16(one, one, None, one, i64(2), ci(8), ci(8), ci(64), ci(64), ci(4), ci(1), cf(0.1), None)
    assert rc < 0 and "kernel size 4" in _err(lib)
    # wgrad without a workspace
    rc = lib.yfs_conv2d_wgrad_bf16(one, one, one, None, i64(2), ci(8), ci(8), ci(64), ci(64), ci(3), ci(1), ci(0), None,
                                   ctypes.c_size_t(0), None)
    assert rc < 0 and "workspace" in _err(lib)
    # fused conv + pool needs an even output
    rc = lib.yfs_conv2d_fwd_pool_bf16(one, one, one, one, one, i64(1), ci(7), ci(7), ci(64), ci(64), ci(3), ci(1), cf(0.1), None)
    assert rc < 0 and "even" in _err(lib)
    # more boxes per image than the NMS kernel holds in LDS
    rc = lib.yfs_nms_classwise_f32(one, one, one, i64(1), ci(5000), cf(0.05), cf(0.5), one, one, None)
    assert rc < 0 and "5000" in _err(lib)
    # loss with 5B+C beyond the per-thread cell buffer
    need = lib.yfs_yolov1_loss_workspace_bytes(i64(1), ci(7))
    rc = lib.yfs_yolov1_loss_fwd_bwd_f32(one, one, i64(1), ci(7), ci(2), ci(4000), cf(5), cf(0.5), one, one, one,
                                         ctypes.c_size_t(need), None)
    assert rc < 0 and "5B+C" in _err(lib)
    # dropout without the mask output, and a Linear step with sizes the tile cannot cover
    rc = lib.yfs_fc_epilogue_fwd_bf16(one, ci(1), None, one, None, i64(4), ci(64), cf(0.1), cf(0.2), ctypes.c_uint64(1), None)
    assert rc < 0 and "keep" in _err(lib)
    rc = lib.yfs_linear_wgrad_sgd_bf16(one, one, i64(8), ci(100), ci(128), one, one, None, None, cf(0), cf(0.1), cf(0.9), cf(0),
                                       ci(0), cf(1), None)
    assert rc < 0 and "multiples of 128" in _err(lib)
    # a success path that launches nothing: workspace size queries
    assert lib.yfs_conv2d_wgrad_workspace_bytes(i64(64), ci(56), ci(56), ci(256), ci(512), ci(3), ci(1)) > 0
    assert lib.yfs_conv2d_wgrad_workspace_bytes(i64(0), ci(56), ci(56), ci(256), ci(512), ci(3), ci(1)) == 0
