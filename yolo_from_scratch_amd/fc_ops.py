"""Host side of the fused fc.1 epilogue (include/yfs.h: yfs_fc_epilogue_{fwd,bwd}_bf16).
Reference: yolo_toy_regressor_torch.py:258-263  Linear -> LeakyReLU(0.1) -> Dropout(0.2)."""
import ctypes

import torch

from . import _lib as L


def fc_epilogue_fwd(partial, bias, slope=0.1, p_drop=0.0, seed=0):
    """partial fp32 [S, M, N] (split-K slabs of the GEMM) or [M, N]; bias fp32 [N] or None.
    -> (y bf16 [M, N] = dropout(leaky(sum_s partial + bias)), keep uint8 [M, N] or None when p_drop == 0)."""
    L.require_cuda(partial, bias)
    if partial.dim() == 2:
        partial = partial.unsqueeze(0)
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise L.YfsError("fc_epilogue_fwd: partial must be contiguous fp32 [S, M, N]")
    s, m, n = partial.shape
    y = torch.empty(m, n, device=partial.device, dtype=torch.bfloat16)
    keep = torch.empty(m, n, device=partial.device, dtype=torch.uint8) if p_drop > 0 else None
    b = bias.detach().float().contiguous() if bias is not None else None
    L.check(L.lib().yfs_fc_epilogue_fwd_bf16(L.ptr(partial), L.c_int(s), L.ptr(b), L.ptr(y), L.ptr(keep), L.i64(m), L.c_int(n),
                                             L.c_float(slope), L.c_float(p_drop), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                             L.stream()), "yfs_fc_epilogue_fwd_bf16")
    return y, keep


def fc_epilogue_bwd(dy, y, keep, slope=0.1, p_drop=0.0, db=None, accumulate=False):
    """-> dz bf16 [M, N]; the bias gradient is written (or added) into ``db`` fp32 [N] when given."""
    L.require_cuda(dy, y, keep, db)
    dy = dy.to(torch.bfloat16).contiguous()
    m, n = y.shape
    dz = torch.empty_like(y)
    L.check(L.lib().yfs_fc_epilogue_bwd_bf16(L.ptr(dy), L.ptr(y), L.ptr(keep), L.ptr(dz), L.ptr(db), L.i64(m), L.c_int(n),
                                             L.c_float(slope), L.c_float(p_drop), L.c_int(1 if accumulate else 0), L.stream()),
            "yfs_fc_epilogue_bwd_bf16")
    return dz


def linear_wgrad_sgd(x16, dy16, p, m, p16, sumsq, max_norm, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    """Weight gradient of a Linear layer fused with its SGD step (include/yfs.h: yfs_linear_wgrad_sgd_bf16): p [Out,In]
    fp32, m (momentum) and p16 (bf16 shadow, may be None) are updated in place from x16 [B,In], dy16 [B,Out]; ``sumsq``
    = device scalar with the whole model's squared gradient norm (None = no clipping)."""
    L.require_cuda(x16, dy16, p, m)
    if x16.dtype != torch.bfloat16 or dy16.dtype != torch.bfloat16 or not (x16.is_contiguous() and dy16.is_contiguous()):
        raise L.YfsError("linear_wgrad_sgd: x and dy must be contiguous bf16")
    b, n_in = x16.shape
    n_out = dy16.shape[1]
    if dy16.shape[0] != b or tuple(p.shape) != (n_out, n_in) or not p.is_contiguous() or p.dtype != torch.float32:
        raise L.YfsError("linear_wgrad_sgd: shape / layout mismatch")
    L.check(L.lib().yfs_linear_wgrad_sgd_bf16(L.ptr(x16), L.ptr(dy16), L.i64(b), L.c_int(n_in), L.c_int(n_out), L.ptr(p), L.ptr(m),
                                              L.ptr(p16), L.ptr(sumsq), L.c_float(max_norm or 0.0), L.c_float(lr),
                                              L.c_float(momentum), L.c_float(weight_decay), L.c_int(1 if first_step else 0),
                                              L.c_float(grad_scale), L.stream()), "yfs_linear_wgrad_sgd_bf16")
