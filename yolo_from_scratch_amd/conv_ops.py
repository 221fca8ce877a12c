"""Functional wrappers over the conv kernels of libyfs.so (csrc/conv_kernels.hip, csrc/pool_kernels.hip).

Tensor convention: activations are torch tensors of logical shape [N,C,H,W] in ``channels_last`` memory
format (physically NHWC), dtype bf16; weights are logical [Cout,Cin,K,K] channels_last (physically
[Cout][K][K][Cin]).  The kernels only ever see the physical layout.
"""
import torch

from . import _lib as L

CL = torch.channels_last
_ws_cache = {}


def nhwc(t, dtype=None):
    """Return t (optionally cast) as a channels_last-dense tensor; no copy when it already is."""
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.dim() != 4:
        raise L.YfsError(f"expected a 4-d NCHW tensor, got {tuple(t.shape)}")
    if not t.is_contiguous(memory_format=CL):
        t = t.contiguous(memory_format=CL)
    return t


def _empty_nhwc(n, c, h, w, device, dtype=torch.bfloat16):
    return torch.empty((n, c, h, w), device=device, dtype=dtype, memory_format=CL)


def out_hw(h, w, k, stride):
    pad = k // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def workspace(nbytes, device):
    """Per-device scratch reused across calls on the current stream (caller-owned per the C ABI)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def weight_prep(w, want_w16=True, want_wt16=True):
    """fp32 weight (logical [Cout,Cin,K,K], channels_last) -> bf16 copy and transposed+flipped bf16 copy."""
    L.require_cuda(w)
    w = nhwc(w, torch.float32)
    cout, cin, k, _ = w.shape
    w16 = torch.empty_like(w, dtype=torch.bfloat16) if want_w16 else None           # [Cout][K][K][Cin]
    wt16 = _empty_nhwc(cin, cout, k, k, w.device) if want_wt16 else None              # [Cin][K][K][Cout]
    L.check(L.lib().yfs_weight_prep_bf16(L.ptr(w), L.ptr(w16), L.ptr(wt16), L.c_int(cout), L.c_int(k), L.c_int(cin),
                                         L.stream()), "yfs_weight_prep_bf16")
    return w16, wt16


def conv_fwd(x, w16, bias, stride=1, slope=0.1):
    """leaky(conv(x, w) + bias).  x bf16 NHWC; w16 bf16 [Cout][K][K][Cin]; bias fp32."""
    L.require_cuda(x, w16, bias)
    x = nhwc(x, torch.bfloat16)
    n, cin, h, w = x.shape
    cout, cin2, k, _ = w16.shape
    if cin2 != cin:
        raise L.YfsError(f"conv_fwd: input has {cin} channels, weight expects {cin2}")
    oh, ow = out_hw(h, w, k, stride)
    y = _empty_nhwc(n, cout, oh, ow, x.device)
    L.check(L.lib().yfs_conv2d_fwd_bf16(L.ptr(x), L.ptr(w16), L.ptr(bias), L.ptr(y), L.i64(n), L.c_int(h), L.c_int(w),
                                        L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride), L.c_float(slope),
                                        L.stream()), "yfs_conv2d_fwd_bf16")
    return y


def conv_dgrad(dz, wt16, in_hw, stride=1, x_act=None, slope=0.1):
    """dx = conv_transpose(dz, w) [* leaky'(x_act)].  wt16 bf16 [Cin][K][K][Cout] from weight_prep."""
    L.require_cuda(dz, wt16, x_act)
    dz = nhwc(dz, torch.bfloat16)
    n, cout = dz.shape[0], dz.shape[1]
    cin, cout2, k, _ = wt16.shape
    h, w = in_hw
    if cout2 != cout or tuple(dz.shape[2:]) != out_hw(h, w, k, stride):
        raise L.YfsError(f"conv_dgrad: dz {tuple(dz.shape)} does not match weight/input geometry")
    dx = _empty_nhwc(n, cin, h, w, dz.device)
    if x_act is not None:
        x_act = nhwc(x_act, torch.bfloat16)
    L.check(L.lib().yfs_conv2d_dgrad_bf16(L.ptr(dz), L.ptr(wt16), L.ptr(x_act), L.ptr(dx), L.i64(n), L.c_int(h),
                                          L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                          L.c_float(slope), L.stream()), "yfs_conv2d_dgrad_bf16")
    return dx


def conv_wgrad(x, dz, k, stride=1, dw=None, db=None, want_db=True, accumulate=None):
    """dw fp32 (logical [Cout,Cin,K,K] channels_last), db fp32 [Cout].  When dw/db are given the result is
    written into them: added when ``accumulate`` (default for given buffers), else overwritten."""
    L.require_cuda(x, dz, dw, db)
    x = nhwc(x, torch.bfloat16)
    dz = nhwc(dz, torch.bfloat16)
    n, cin, h, w = x.shape
    cout = dz.shape[1]
    if accumulate is None:
        accumulate = dw is not None
    if dw is None:
        dw = _empty_nhwc(cout, cin, k, k, x.device, torch.float32)
        db = torch.empty(cout, device=x.device, dtype=torch.float32) if want_db else None
    elif not dw.is_contiguous(memory_format=CL) or dw.dtype != torch.float32:
        raise L.YfsError("conv_wgrad: dw must be fp32 channels_last")
    nbytes = L.lib().yfs_conv2d_wgrad_workspace_bytes(L.i64(n), L.c_int(h), L.c_int(w), L.c_int(cin), L.c_int(cout),
                                                      L.c_int(k), L.c_int(stride))
    ws = workspace(nbytes, x.device)
    L.check(L.lib().yfs_conv2d_wgrad_bf16(L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(db), L.i64(n), L.c_int(h), L.c_int(w),
                                          L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                          L.c_int(1 if accumulate else 0), L.ptr(ws), L.c_size(nbytes), L.stream()),
            "yfs_conv2d_wgrad_bf16")
    return dw, db


def leaky_bwd(dy, y, slope=0.1):
    L.require_cuda(dy, y)
    dy = nhwc(dy, torch.bfloat16)
    y = nhwc(y, torch.bfloat16)
    dz = torch.empty_like(dy)
    L.check(L.lib().yfs_leaky_bwd_bf16(L.ptr(dy), L.ptr(y), L.ptr(dz), L.i64(dy.numel()), L.c_float(slope), L.stream()),
            "yfs_leaky_bwd_bf16")
    return dz


def im2col(x, k, stride, kpad):
    """Stem helper: x [N,C,H,W] channels_last (fp32 or bf16) -> cols bf16 logical [N, kpad, OH, OW] channels_last,
    i.e. physically [N*OH*OW, kpad] with (kh, kw, ci) order, zero padded."""
    L.require_cuda(x)
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = nhwc(x)
    n, c, h, w = x.shape
    oh, ow = out_hw(h, w, k, stride)
    cols = _empty_nhwc(n, kpad, oh, ow, x.device)
    L.check(L.lib().yfs_im2col_bf16(L.ptr(x), L.c_int(1 if x.dtype == torch.float32 else 0), L.ptr(cols), L.i64(n),
                                    L.c_int(h), L.c_int(w), L.c_int(c), L.c_int(k), L.c_int(stride), L.c_int(kpad),
                                    L.stream()), "yfs_im2col_bf16")
    return cols
