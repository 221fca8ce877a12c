"""Functional wrappers over the conv kernels of libyfs.so (csrc/conv_kernels.hip, csrc/pool_kernels.hip).

Tensor convention: activations are torch tensors of logical shape [N,C,H,W] in ``channels_last`` memory
format (physically NHWC), dtype bf16; weights are logical [Cout,Cin,K,K] channels_last (physically
[Cout][K][K][Cin]).  The kernels only ever see the physical layout.
"""
import ctypes

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
    """Scratch reused across calls, one buffer per (device, current stream): the backward chain runs weight
    gradients on a second stream, and a buffer shared between streams could be regrown (freed to one stream's pool)
    or rewritten while a kernel of the other stream still uses it.  Caller-owned per the C ABI."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


_split_bytes = {}
SPLIT_K = True      # False: never hand the kernels a split-K workspace (every output is then independent of the batch size)


def _split_workspace(op, n, h, w, cin, cout, k, stride, device):
    """(scratch, bytes) for the split-K form of a forward / dgrad launch: (None, 0) for every shape that does not
    split (all but the few-pixel, long-K layers).  The byte count per shape is cached, but not while a tile is forced
    through the tuning knob (it depends on the knob then)."""
    if not SPLIT_K:
        return None, 0
    key = (op, n, h, w, cin, cout, k, stride)
    nbytes = _split_bytes.get(key)
    if nbytes is None or L.get_knob("igemm_tile") != 0:
        fn = L.lib().yfs_conv2d_fwd_workspace_bytes if op == "fwd" else L.lib().yfs_conv2d_dgrad_workspace_bytes
        nbytes = fn(L.i64(n), L.c_int(h), L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride))
        if L.get_knob("igemm_tile") == 0:
            _split_bytes[key] = nbytes
    return (workspace(nbytes, device), nbytes) if nbytes else (None, 0)


def release_workspaces():
    """Drop the cached scratch buffers of this module and of data.py (one per device and stream that ever launched a
    kernel needing one; they are kept for reuse and otherwise live until the process ends).  Safe between steps: the
    next launch allocates again."""
    from . import data
    _ws_cache.clear()
    data._ws.clear()


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


def conv_fwd_pool(x, w16, bias, stride=1, slope=0.1):
    """max_pool2x2(leaky(conv(x, w) + bias)) in one launch -> (pooled y [N,Cout,OH/2,OW/2] bf16, idx int16): the pool
    runs in the GEMM epilogue and the full-resolution activation is never written."""
    L.require_cuda(x, w16, bias)
    x = nhwc(x, torch.bfloat16)
    n, cin, h, w = x.shape
    cout, cin2, k, _ = w16.shape
    if cin2 != cin:
        raise L.YfsError(f"conv_fwd_pool: input has {cin} channels, weight expects {cin2}")
    oh, ow = out_hw(h, w, k, stride)
    y = _empty_nhwc(n, cout, oh // 2, ow // 2, x.device)
    idx = torch.empty(n * (oh // 2) * (ow // 2) * (cout // 8), device=x.device, dtype=torch.int16)
    L.check(L.lib().yfs_conv2d_fwd_pool_bf16(L.ptr(x), L.ptr(w16), L.ptr(bias), L.ptr(y), L.ptr(idx), L.i64(n), L.c_int(h),
                                             L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                             L.c_float(slope), L.stream()), "yfs_conv2d_fwd_pool_bf16")
    return y, idx


class _WeightDesc(ctypes.Structure):       # include/yfs.h: yfs_weight_desc
    _fields_ = [("w", ctypes.c_void_p), ("w_bf16", ctypes.c_void_p), ("wt_bf16", ctypes.c_void_p),
                ("Cout", ctypes.c_int), ("K", ctypes.c_int), ("Cin", ctypes.c_int), ("w_is_bf16", ctypes.c_int)]


def weight_prep_batch(items):
    """items: list of (w, want_w16, want_wt16) with w the fp32 master weight or its bf16 shadow (logical
    [Cout,Cin,K,K], channels_last).  One launch for all of them; returns a list of (w16, wt16) — w16 is the shadow
    itself when the source already is bf16."""
    out, descs = [], []
    for w, want_w16, want_wt16 in items:
        L.require_cuda(w)
        src16 = w.dtype == torch.bfloat16
        w = nhwc(w, w.dtype if src16 else torch.float32)
        cout, cin, k, _ = w.shape
        w16 = w if src16 else (torch.empty_like(w, dtype=torch.bfloat16) if want_w16 else None)
        wt16 = _empty_nhwc(cin, cout, k, k, w.device) if want_wt16 else None
        out.append((w16, wt16))
        if wt16 is not None or (w16 is not None and not src16):
            descs.append((w, None if src16 else w16, wt16, cout, k, cin, int(src16)))
    for i in range(0, len(descs), 32):
        chunk = descs[i:i + 32]
        arr = (_WeightDesc * len(chunk))()
        for a, (w, w16, wt16, cout, k, cin, s16) in zip(arr, chunk):
            a.w, a.w_bf16, a.wt_bf16 = w.data_ptr(), (w16.data_ptr() if w16 is not None else None), \
                (wt16.data_ptr() if wt16 is not None else None)
            a.Cout, a.K, a.Cin, a.w_is_bf16 = cout, k, cin, s16
        L.check(L.lib().yfs_weight_prep_batch_bf16(arr, L.c_int(len(chunk)), L.stream()), "yfs_weight_prep_batch_bf16")
    return out


def fold_batchnorm(bn, conv_bias=None):
    """(scale, shift) of an eval-mode nn.BatchNorm2d folded behind a convolution: scale = gamma / sqrt(var + eps),
    shift = beta + (conv_bias - mean) * scale (README.md:30 of the reference names Conv + BN + LeakyReLU as its next
    block; the snapshot itself has no BatchNorm)."""
    inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(inv)
    beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(inv)
    scale = gamma * inv
    b = conv_bias.detach().float() if conv_bias is not None else torch.zeros_like(inv)
    return scale.contiguous(), (beta + (b - bn.running_mean.detach().float()) * scale).contiguous()


def conv_fwd_affine(x, w16, scale, shift, stride=1, slope=0.1):
    """leaky(conv(x, w) * scale[c] + shift[c]): Conv + folded BatchNorm + LeakyReLU in one launch (scale=None: 1)."""
    L.require_cuda(x, w16, scale, shift)
    x = nhwc(x, torch.bfloat16)
    n, cin, h, w = x.shape
    cout, cin2, k, _ = w16.shape
    if cin2 != cin:
        raise L.YfsError(f"conv_fwd_affine: input has {cin} channels, weight expects {cin2}")
    oh, ow = out_hw(h, w, k, stride)
    y = _empty_nhwc(n, cout, oh, ow, x.device)
    ws, nbytes = _split_workspace("fwd", n, h, w, cin, cout, k, stride, x.device)
    L.check(L.lib().yfs_conv2d_fwd_ws_bf16(L.ptr(x), L.ptr(w16), L.ptr(scale), L.ptr(shift), L.ptr(y), L.i64(n), L.c_int(h),
                                           L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                           L.c_float(slope), L.ptr(ws), L.c_size(nbytes), L.stream()), "yfs_conv2d_fwd_ws_bf16")
    return y


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
    ws, nbytes = _split_workspace("fwd", n, h, w, cin, cout, k, stride, x.device)
    L.check(L.lib().yfs_conv2d_fwd_ws_bf16(L.ptr(x), L.ptr(w16), L.ptr(None), L.ptr(bias), L.ptr(y), L.i64(n), L.c_int(h),
                                           L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                           L.c_float(slope), L.ptr(ws), L.c_size(nbytes), L.stream()), "yfs_conv2d_fwd_ws_bf16")
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
    ws, nbytes = _split_workspace("dgrad", n, h, w, cin, cout, k, stride, dz.device)
    L.check(L.lib().yfs_conv2d_dgrad_ws_bf16(L.ptr(dz), L.ptr(wt16), L.ptr(x_act), L.ptr(dx), L.i64(n), L.c_int(h),
                                             L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k), L.c_int(stride),
                                             L.c_float(slope), L.ptr(ws), L.c_size(nbytes), L.stream()),
            "yfs_conv2d_dgrad_ws_bf16")
    return dx


def conv_dgrad_unpool(dz, wt16, y_pooled, pool_idx, slope=0.1):
    """Input gradient of a stride-1 block whose input is the output of a 2x2 max pool, carried through that pool in the
    same launch: returns the gradient of the pool's INPUT [N, Cin, 2H, 2W] (the LeakyReLU-masked dgrad value at the
    window position ``pool_idx`` recorded, zeros at the other three).  Bit-identical to
    ``conv_dgrad(..., x_act=y_pooled)`` followed by the pool backward; the pooled-resolution gradient is never written."""
    L.require_cuda(dz, wt16, y_pooled, pool_idx)
    dz = nhwc(dz, torch.bfloat16)
    y_pooled = nhwc(y_pooled, torch.bfloat16)
    n, cout = dz.shape[0], dz.shape[1]
    cin, cout2, k, _ = wt16.shape
    h, w = y_pooled.shape[2:]
    if cout2 != cout or tuple(dz.shape[2:]) != (h, w) or y_pooled.shape[1] != cin or y_pooled.shape[0] != n:
        raise L.YfsError(f"conv_dgrad_unpool: dz {tuple(dz.shape)} / y_pooled {tuple(y_pooled.shape)} do not match the weight")
    if pool_idx.dtype != torch.int16 or pool_idx.numel() != n * h * w * (cin // 8):
        raise L.YfsError("conv_dgrad_unpool: pool_idx must be the int16 record of the pool that produced y_pooled")
    dx = _empty_nhwc(n, cin, 2 * h, 2 * w, dz.device)
    L.check(L.lib().yfs_conv2d_dgrad_unpool_bf16(L.ptr(dz), L.ptr(wt16), L.ptr(y_pooled), L.ptr(pool_idx), L.ptr(dx), L.i64(n),
                                                 L.c_int(h), L.c_int(w), L.c_int(cin), L.c_int(cout), L.c_int(k),
                                                 L.c_float(slope), L.stream()), "yfs_conv2d_dgrad_unpool_bf16")
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


# ---- stem: Conv2D(3, Cout, 7, s=2) through the 2x2 space-to-depth image (no im2col matrix) ----------------------
_STEM_IDX = {}


def stem_geometry(h, w):
    oh, ow = out_hw(h, w, 7, 2)
    return oh, ow, oh + 4, ow + 4


def stem_space_to_depth(x):
    """x [N,3,H,W] (fp32 or bf16, any strides) -> xs bf16, logical [N,16,Hs,Ws] channels_last."""
    L.require_cuda(x)
    if x.dim() != 4 or x.shape[1] != 3:
        raise L.YfsError(f"stem expects [N,3,H,W], got {tuple(x.shape)}")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    n, _, h, w = x.shape
    _, _, hs, ws = stem_geometry(h, w)
    xs = _empty_nhwc(n, 16, hs, ws, x.device)
    sn, sc, sh, sw = x.stride()
    L.check(L.lib().yfs_stem_space_to_depth_bf16(L.ptr(x), L.c_int(1 if x.dtype == torch.float32 else 0), L.i64(sn),
                                                 L.i64(sc), L.i64(sh), L.i64(sw), L.ptr(xs), L.i64(n), L.c_int(h),
                                                 L.c_int(w), L.c_int(hs), L.c_int(ws), L.stream()),
            "yfs_stem_space_to_depth_bf16")
    return xs


def _stem_index(device):
    """Flat gather index: w2.view(Cout, 256)[:, k] = w_nhwc.view(Cout, 147)[:, idx[k]] where mask[k]."""
    key = str(device)
    if key not in _STEM_IDX:
        idx = torch.zeros(4, 4, 16, dtype=torch.long)
        mask = torch.zeros(4, 4, 16, dtype=torch.bool)
        for a in range(4):
            for b in range(4):
                for dy in range(2):
                    for dx in range(2):
                        kh, kw = 2 * a + dy, 2 * b + dx
                        if kh > 6 or kw > 6:
                            continue
                        for c in range(3):
                            e = (dy * 2 + dx) * 3 + c
                            idx[a, b, e] = (kh * 7 + kw) * 3 + c       # position in [7][7][3]
                            mask[a, b, e] = True
        _STEM_IDX[key] = (idx.view(-1).to(device), mask.view(-1).to(device))
    return _STEM_IDX[key]


def stem_weight(w):
    """fp32 weight logical [Cout,3,7,7] channels_last -> w2 bf16 [Cout, 4*64] (see include/yfs.h)."""
    cout = w.shape[0]
    idx, mask = _stem_index(w.device)
    flat = w.detach().permute(0, 2, 3, 1).reshape(cout, 147)
    return (flat[:, idx] * mask).to(torch.bfloat16).contiguous()


def _stem_inverse(device):
    """inv[j] = the one position k of w2's 256 columns that holds weight j of [7][7][3] (built once per device:
    boolean-mask indexing here would run nonzero() and stall the host on the stream at every step)."""
    key = ("inv", str(device))
    if key not in _STEM_IDX:
        idx, mask = _stem_index(device)
        idx, mask = idx.cpu(), mask.cpu()
        inv = torch.full((147,), -1, dtype=torch.long)
        inv[idx[mask]] = torch.arange(256)[mask]
        assert int(inv.min()) >= 0
        _STEM_IDX[key] = inv.to(device)
    return _STEM_IDX[key]


def stem_weight_grad(dw2, cout):
    """dw2 fp32 [Cout, 256] -> gradient in [Cout, 147] ((kh,kw,c) order): every real weight appears once in w2."""
    return dw2.view(cout, 256)[:, _stem_inverse(dw2.device)]


def stem_fwd(xs, w2, bias, oh, ow, slope=0.1):
    n, _, hs, ws = xs.shape
    cout = w2.shape[0]
    y = _empty_nhwc(n, cout, oh, ow, xs.device)
    L.check(L.lib().yfs_conv2d_stem_fwd_bf16(L.ptr(xs), L.ptr(w2), L.ptr(bias), L.ptr(y), L.i64(n), L.c_int(hs), L.c_int(ws),
                                             L.c_int(oh), L.c_int(ow), L.c_int(cout), L.c_float(slope), L.stream()),
            "yfs_conv2d_stem_fwd_bf16")
    return y


def stem_fwd_pool(xs, w2, bias, oh, ow, slope=0.1):
    """Stem conv + LeakyReLU + 2x2 max pool in one launch -> (pooled y, idx); see conv_fwd_pool."""
    n, _, hs, ws = xs.shape
    cout = w2.shape[0]
    y = _empty_nhwc(n, cout, oh // 2, ow // 2, xs.device)
    idx = torch.empty(n * (oh // 2) * (ow // 2) * (cout // 8), device=xs.device, dtype=torch.int16)
    L.check(L.lib().yfs_conv2d_stem_fwd_pool_bf16(L.ptr(xs), L.ptr(w2), L.ptr(bias), L.ptr(y), L.ptr(idx), L.i64(n),
                                                  L.c_int(hs), L.c_int(ws), L.c_int(oh), L.c_int(ow), L.c_int(cout),
                                                  L.c_float(slope), L.stream()), "yfs_conv2d_stem_fwd_pool_bf16")
    return y, idx


def stem_wgrad(xs, dz):
    """-> (dw as [Cout,147] fp32 in (kh,kw,c) order, db [Cout])."""
    dz = nhwc(dz, torch.bfloat16)
    n, _, hs, ws = xs.shape
    cout, oh, ow = dz.shape[1], dz.shape[2], dz.shape[3]
    dw2 = torch.empty(cout, 256, device=xs.device, dtype=torch.float32)
    db = torch.empty(cout, device=xs.device, dtype=torch.float32)
    nbytes = L.lib().yfs_conv2d_stem_wgrad_workspace_bytes(L.i64(n), L.c_int(oh), L.c_int(ow), L.c_int(cout))
    ws_buf = workspace(nbytes, xs.device)
    L.check(L.lib().yfs_conv2d_stem_wgrad_bf16(L.ptr(xs), L.ptr(dz), L.ptr(dw2), L.ptr(db), L.i64(n), L.c_int(hs),
                                               L.c_int(ws), L.c_int(oh), L.c_int(ow), L.c_int(cout), L.c_int(0),
                                               L.ptr(ws_buf), L.c_size(nbytes), L.stream()), "yfs_conv2d_stem_wgrad_bf16")
    return stem_weight_grad(dw2, cout), db


def stem_wgrad_pool(xs, dy_pooled, idx, y_pooled, slope=0.1):
    """stem_wgrad for a stem whose output went through LeakyReLU and the fused 2x2 pool, from the pooled side:
    (pooled gradient, argmax record, pooled activation) instead of a full-resolution dz.  Same return."""
    dy_pooled = nhwc(dy_pooled, torch.bfloat16)
    n, _, hs, ws = xs.shape
    cout, oh, ow = dy_pooled.shape[1], 2 * dy_pooled.shape[2], 2 * dy_pooled.shape[3]
    dw2 = torch.empty(cout, 256, device=xs.device, dtype=torch.float32)
    db = torch.empty(cout, device=xs.device, dtype=torch.float32)
    nbytes = L.lib().yfs_conv2d_stem_wgrad_workspace_bytes(L.i64(n), L.c_int(oh), L.c_int(ow), L.c_int(cout))
    ws_buf = workspace(nbytes, xs.device)
    L.check(L.lib().yfs_conv2d_stem_wgrad_pool_bf16(L.ptr(xs), L.ptr(dy_pooled), L.ptr(idx), L.ptr(y_pooled), L.ptr(dw2),
                                                    L.ptr(db), L.i64(n), L.c_int(hs), L.c_int(ws), L.c_int(oh), L.c_int(ow),
                                                    L.c_int(cout), L.c_float(slope), L.c_int(0), L.ptr(ws_buf),
                                                    L.c_size(nbytes), L.stream()), "yfs_conv2d_stem_wgrad_pool_bf16")
    return stem_weight_grad(dw2, cout), db
