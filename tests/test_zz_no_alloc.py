"""yfs_yolov1_loss_noalloc_f32 (the entry point that allocates nothing, include/yfs.h) against
yfs_yolov1_loss_fwd_bwd_f32, through the C ABI: two calls in a row, the second with a smaller input, must give
the workspace form's loss terms and gradient bit for bit (same kernels, same reduction order)."""
import ctypes

import numpy as np
import pytest
import torch

from yolo_from_scratch_amd import _lib as L
from yolo_from_scratch_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(N, seed):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.2, 1.2, (N, 7, 7, 30)).astype(np.float32)
    t = synth.v1_targets(N, seed=seed + 1)
    return torch.from_numpy(p).to(DEV), torch.from_numpy(t).to(DEV)


def _workspace_form(p, t):
    N = p.shape[0]
    nb = L.lib().yfs_yolov1_loss_workspace_bytes(L.i64(N), L.c_int(7))
    ws = torch.empty(nb // 4 + 1, device=DEV, dtype=torch.float32)
    out = torch.full((5,), -1.0, device=DEV)
    grad = torch.full_like(p, 7.0)
    L.check(L.lib().yfs_yolov1_loss_fwd_bwd_f32(L.ptr(p), L.ptr(t), L.i64(N), L.c_int(7), L.c_int(2), L.c_int(20),
                                                L.c_float(5.0), L.c_float(0.5), L.ptr(out), L.ptr(grad), L.ptr(ws),
                                                L.c_size(nb), L.stream()), "yfs_yolov1_loss_fwd_bwd_f32")
    return out, grad


def _noalloc_form(p, t):
    N = p.shape[0]
    out = torch.full((5,), -2.0, device=DEV)
    grad = torch.full_like(p, 9.0)
    L.check(L.lib().yfs_yolov1_loss_noalloc_f32(L.ptr(p), L.ptr(t), L.i64(N), L.c_int(7), L.c_int(2), L.c_int(20),
                                                L.c_float(5.0), L.c_float(0.5), L.ptr(out), L.ptr(grad), L.stream()),
            "yfs_yolov1_loss_noalloc_f32")
    return out, grad


def test_noalloc_matches_workspace_form_twice_second_smaller():
    big, small = _inputs(700, 3), _inputs(37, 11)
    refs = [_workspace_form(*big), _workspace_form(*small)]
    got = [_noalloc_form(*big), _noalloc_form(*small)]       # back to back: the second reuses the same scratch
    torch.cuda.synchronize()
    for (ro, rg), (go, gg) in zip(refs, got):
        assert torch.equal(ro, go), (ro, go)
        assert torch.equal(rg, gg)


def test_noalloc_rejects_oversize():
    lib = L.lib()
    lib.yfs_yolov1_loss_noalloc_max_cells.restype = ctypes.c_int64
    cap = lib.yfs_yolov1_loss_noalloc_max_cells()
    assert cap == 65536 * 256
    one = ctypes.c_void_p(16)
    rc = lib.yfs_yolov1_loss_noalloc_f32(one, one, L.i64(cap // 49 + 1), L.c_int(7), L.c_int(2), L.c_int(20), L.c_float(5.0),
                                         L.c_float(0.5), one, one, None)
    assert rc == -2 and "exceed" in L.last_error()
