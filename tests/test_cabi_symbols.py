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
    assert lib.yfs_version() >= 100
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
