"""Drop-in model surface of the reference on the MI355X kernels.

Mirrors yolo_toy_regressor_torch.py: ``Conv2D(in_ch, out_ch, k, s=1)`` (:206-211) and
``YoloRegressorModel`` (:213-269) with the same attribute names (``feat``, ``fc``), the same module
indices and therefore the same state-dict keys (``feat.<i>.0.weight/bias``, ``fc.1.*``, ``fc.4.*``), so a
``best_val_iou.pt`` written by the reference loads here and vice versa.

What differs underneath: activations are bf16 NHWC, every Conv2D block runs as one fused MFMA
implicit-GEMM launch (conv + bias + LeakyReLU), and when ``feat`` is called as a whole the backward
pass runs as one chain in which each dgrad applies the previous block's LeakyReLU derivative in its
epilogue and each pool backward does the same for the block feeding it.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import conv_ops as co
from . import fc_ops as fo

SLOPE = 0.1
_TWO_STREAMS = os.environ.get("YFS_WGRAD_STREAM", "1") != "0"
_FUSE_POOL = os.environ.get("YFS_FUSE_POOL", "1") != "0"
_FUSE_STEM_POOL_BWD = os.environ.get("YFS_FUSE_STEM_POOL_BWD", "1") != "0"
# pool backward inside the dgrad of the block behind it (conv_ops.conv_dgrad_unpool): bit-identical, and measured no faster
# than the two launches inside the step (10.62 vs 10.63 ms, same box), so it is opt-in
_FUSE_UNPOOL = os.environ.get("YFS_FUSE_UNPOOL", "0") == "1"
_FUSE_FC = os.environ.get("YFS_FUSE_FC", "1") != "0"


def _pool_fwd(x):
    x = co.nhwc(x, torch.bfloat16)
    n, c, h, w = x.shape
    y = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.bfloat16, memory_format=co.CL)
    L.check(L.lib().yfs_maxpool2_fwd_bf16(L.ptr(x), L.ptr(y), L.i64(n), L.c_int(h), L.c_int(w), L.c_int(c), L.stream()),
            "yfs_maxpool2_fwd_bf16")
    return y


def _pool_fwd_idx(x):
    """Pool forward that also saves the argmax positions (2 bits per element) for _pool_bwd_idx."""
    x = co.nhwc(x, torch.bfloat16)
    n, c, h, w = x.shape
    y = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.bfloat16, memory_format=co.CL)
    idx = torch.empty(n * (h // 2) * (w // 2) * (c // 8), device=x.device, dtype=torch.int16)
    L.check(L.lib().yfs_maxpool2_fwd_idx_bf16(L.ptr(x), L.ptr(y), L.ptr(idx), L.i64(n), L.c_int(h), L.c_int(w), L.c_int(c),
                                              L.stream()), "yfs_maxpool2_fwd_idx_bf16")
    return y, idx


def _pool_bwd_idx(y, idx, dy, fuse_leaky, slope=SLOPE):
    dy = co.nhwc(dy, torch.bfloat16)
    n, c, ho, wo = y.shape
    dx = torch.empty((n, c, 2 * ho, 2 * wo), device=y.device, dtype=torch.bfloat16, memory_format=co.CL)
    L.check(L.lib().yfs_maxpool2_bwd_idx_bf16(L.ptr(y), L.ptr(idx), L.ptr(dy), L.ptr(dx), L.i64(n), L.c_int(2 * ho),
                                              L.c_int(2 * wo), L.c_int(c), L.c_int(1 if fuse_leaky else 0),
                                              L.c_float(slope), L.stream()), "yfs_maxpool2_bwd_idx_bf16")
    return dx


def _pool_bwd(x, dy, fuse_leaky, slope=SLOPE):
    x = co.nhwc(x, torch.bfloat16)
    dy = co.nhwc(dy, torch.bfloat16)
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    L.check(L.lib().yfs_maxpool2_bwd_bf16(L.ptr(x), L.ptr(dy), L.ptr(dx), L.i64(n), L.c_int(h), L.c_int(w), L.c_int(c),
                                          L.c_int(1 if fuse_leaky else 0), L.c_float(slope), L.stream()),
            "yfs_maxpool2_bwd_bf16")
    return dx


class _ConvPlan:
    """What one Conv2D block needs in forward and backward; shared by the per-layer and the chained path."""

    def __init__(self, weight, bias, stride, need_wt, prepared=None):
        cout, cin, k, _ = weight.shape
        self.k, self.stride, self.cin, self.cout = k, stride, cin, cout
        self.stem = cin % 64 != 0
        if self.stem:
            if (cin, k, stride) != (3, 7, 2):
                raise L.YfsError(f"Conv2D({cin}, {cout}, {k}, s={stride}): Cin must be a multiple of 64 except for "
                                 "the 3-channel 7x7/s2 stem")
            self.w16 = co.stem_weight(weight)
            self.wt16 = None
        else:
            self.w16, self.wt16 = prepared if prepared is not None else self.prepare([(weight, need_wt)])[0]
        self.bias = bias.detach().float().contiguous()

    @staticmethod
    def prepare(weights):
        """bf16 operand copies for a list of (weight, need_wt) in one launch: [Cout][K][K][Cin] (the optimiser's
        bf16 shadow when there is one) and, if asked, the transposed tap-reversed copy dgrad reads."""
        from .train import bf16_shadow
        items = []
        for w, need_wt in weights:
            sh = bf16_shadow(w)                 # written by the optimiser step; None without FlatSGD
            items.append((sh if sh is not None else w.detach(), True, need_wt))
        return co.weight_prep_batch(items)

    def forward(self, x):
        """x: NHWC bf16 (or fp32/bf16 image for the stem). Returns (gemm_input, y)."""
        if self.stem:
            xs = co.stem_space_to_depth(x)
            oh, ow, _, _ = co.stem_geometry(x.shape[2], x.shape[3])
            return xs, co.stem_fwd(xs, self.w16, self.bias, oh, ow, SLOPE)
        x = co.nhwc(x, torch.bfloat16)
        return x, co.conv_fwd(x, self.w16, self.bias, self.stride, SLOPE)

    def forward_pool(self, x):
        """Block + the 2x2 max pool that follows it, fused.  Returns (gemm_input, pooled y, idx)."""
        if self.stem:
            xs = co.stem_space_to_depth(x)
            oh, ow, _, _ = co.stem_geometry(x.shape[2], x.shape[3])
            return (xs,) + co.stem_fwd_pool(xs, self.w16, self.bias, oh, ow, SLOPE)
        x = co.nhwc(x, torch.bfloat16)
        return (x,) + co.conv_fwd_pool(x, self.w16, self.bias, self.stride, SLOPE)

    def out_hw(self, h, w):
        if self.stem:
            oh, ow, _, _ = co.stem_geometry(h, w)
            return oh, ow
        return co.out_hw(h, w, self.k, self.stride)

    def wgrad(self, gemm_in, dz, weight, bias, pooled=None):
        """Weight / bias gradients: written into the parameters' gradient sinks when they have one (returns
        None then), else returned (dw in the weight's logical shape, channels_last fp32).
        ``pooled`` = (idx, y_pooled): ``dz`` is then the gradient of the fused pool's OUTPUT (stem only)."""
        from .train import write_grad
        wsink = getattr(weight, "_yfs_grad_sink", None)
        bsink = getattr(bias, "_yfs_grad_sink", None)
        if self.stem:
            if pooled is not None:
                dwv, db = co.stem_wgrad_pool(gemm_in, dz, pooled[0], pooled[1], SLOPE)
            else:
                dwv, db = co.stem_wgrad(gemm_in, dz)       # [Cout, 147] in (kh, kw, c) order
            cout, cin, k, _ = weight.shape

            def put_w(sink, acc):
                if sink is None:
                    return dwv.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
                flat = sink.permute(0, 2, 3, 1).reshape(cout, cin * k * k)   # a view: sink is channels_last
                flat.add_(dwv) if acc else flat.copy_(dwv)

            def put_b(sink, acc):
                if sink is None:
                    return db
                sink.add_(db) if acc else sink.copy_(db)
            return write_grad(weight, put_w), write_grad(bias, put_b)
        if wsink is not None and bsink is not None:
            acc = not weight._yfs_grad_fresh
            co.conv_wgrad(gemm_in, dz, self.k, self.stride, dw=wsink, db=bsink, accumulate=acc)
            weight._yfs_grad_fresh = bias._yfs_grad_fresh = False
            from .train import grad_ready
            grad_ready(weight)
            grad_ready(bias)
            return None, None
        return co.conv_wgrad(gemm_in, dz, self.k, self.stride)


class _ConvBlockFn(torch.autograd.Function):
    """One Conv2D block on its own (used when a block is called outside a Backbone chain)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        L.require_cuda(x, weight, bias)
        plan = _ConvPlan(weight, bias, stride, need_wt=ctx.needs_input_grad[0])
        gin, y = plan.forward(x)
        ctx.plan, ctx.in_shape, ctx.in_dtype = plan, x.shape, x.dtype
        ctx.wb = (weight, bias)
        ctx.save_for_backward(gin, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        gin, y = ctx.saved_tensors
        plan = ctx.plan
        dz = co.leaky_bwd(dy, y, SLOPE)
        dw, db = plan.wgrad(gin, dz, *ctx.wb)
        dx = None
        if ctx.needs_input_grad[0]:
            if plan.stem:
                raise L.YfsError("input gradient of the Cin=3 stem is not implemented (the reference never needs it)")
            dx = co.conv_dgrad(dz, plan.wt16, ctx.in_shape[2:], plan.stride).to(ctx.in_dtype)
        return dx, dw, db, None


class HipConv2d(nn.Module):
    """nn.Conv2d(in_ch, out_ch, k, stride=s, padding=k//2) whose forward also applies LeakyReLU(0.1).
    Parameters keep nn.Conv2d's names, shapes and default init; the weight is stored channels_last."""

    def __init__(self, in_ch, out_ch, k, stride=1):
        super().__init__()
        ref = nn.Conv2d(in_ch, out_ch, k, stride=stride, padding=k // 2)    # default init, same RNG consumption
        self.weight = nn.Parameter(ref.weight.detach().contiguous(memory_format=co.CL))
        self.bias = nn.Parameter(ref.bias.detach())
        self.weight._yfs_sink_capable = self.bias._yfs_sink_capable = True
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_ch, out_ch, k, stride

    def forward(self, x):
        return _ConvBlockFn.apply(x, self.weight, self.bias, self.stride)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, fused LeakyReLU({SLOPE})"


class FusedLeakyReLU(nn.Module):
    """Placeholder at index 1 of a Conv2D block: the activation already ran in the conv epilogue."""

    def __init__(self, negative_slope=SLOPE):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return x


def Conv2D(in_ch, out_ch, k, s=1):
    """Reference signature (yolo_toy_regressor_torch.py:206): Sequential(conv, LeakyReLU(0.1))."""
    return nn.Sequential(HipConv2d(in_ch, out_ch, k, s), FusedLeakyReLU(SLOPE))


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L.require_cuda(x)
        x16 = co.nhwc(x, torch.bfloat16)
        ctx.save_for_backward(x16)
        ctx.in_dtype = x.dtype
        return _pool_fwd(x16)

    @staticmethod
    def backward(ctx, dy):
        (x16,) = ctx.saved_tensors
        return _pool_bwd(x16, dy, fuse_leaky=False).to(ctx.in_dtype)


class HipMaxPool2d(nn.Module):
    """nn.MaxPool2d(kernel_size=2, stride=2) on NHWC bf16."""

    def forward(self, x):
        return _PoolFn.apply(x)


_SIDE = {}


def _side_stream(device):
    """Second HIP stream per device for the weight-gradient kernels of the backward chain."""
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
        from . import dp
        dp.register_producer_stream(_SIDE[key])
    return _SIDE[key]


class _BackboneFn(torch.autograd.Function):
    """All of ``feat`` as one autograd node: forward = chain of fused conv / pool launches; backward walks
    it in reverse with the LeakyReLU derivative folded into dgrad epilogues and pool backward."""

    @staticmethod
    def forward(ctx, x, stages, *params):
        L.require_cuda(x)
        plans, acts, gins = [], [], []
        h = x
        # operand copies of every non-stem layer in one launch
        convs = [i for i, st in enumerate(stages) if st != "P"]
        want_bwd = any(ctx.needs_input_grad)      # all False under torch.no_grad(): no transposed copies for dgrad then
        todo = [(params[2 * j], convs[j] > 0 and want_bwd) for j in range(len(convs)) if params[2 * j].shape[1] % 64 == 0]
        prepared = iter(_ConvPlan.prepare(todo))
        pi = 0
        fused_idx = None
        for si, st in enumerate(stages):
            if st == "P":
                if fused_idx is not None:         # the block before it already pooled in its epilogue
                    idx, fused_idx = fused_idx, None
                else:
                    h, idx = _pool_fwd_idx(h)
                plans.append(None)
                gins.append(idx)          # a pool's backward needs its output and these 2-bit positions only
                if acts:
                    acts[-1] = None       # ... so the full-resolution block output before it is not kept
            else:
                w, b = params[pi], params[pi + 1]
                pi += 2
                plan = _ConvPlan(w, b, st, need_wt=len(plans) > 0,
                                 prepared=next(prepared) if w.shape[1] % 64 == 0 else None)
                oh, ow = plan.out_hw(h.shape[2], h.shape[3])
                if _FUSE_POOL and si + 1 < len(stages) and stages[si + 1] == "P" and oh % 2 == 0 and ow % 2 == 0:
                    gin, h, fused_idx = plan.forward_pool(h)   # h is already the pooled tensor
                else:
                    gin, h = plan.forward(h)
                plans.append(plan)
                gins.append(gin)
            acts.append(h)
        ctx.stages, ctx.plans = stages, plans
        ctx.params = params
        ctx.acts, ctx.gins = acts, gins   # kept on ctx: bf16 activations are internal, never user-visible
        return h

    @staticmethod
    def backward(ctx, dy):
        stages, plans, acts, gins = ctx.stages, ctx.plans, ctx.acts, ctx.gins
        grads = [None] * len(ctx.params)
        nconv = sum(1 for s in stages if s != "P")
        pi = 2 * nconv
        last = len(stages) - 1
        g = co.nhwc(dy, torch.bfloat16)
        if stages[last] != "P":
            g = co.leaky_bwd(g, acts[last], SLOPE)
        # dz of a block feeds two independent kernels: its weight gradient and the next dgrad.  The weight
        # gradients run on a second stream so that the tails of one kernel family (partly filled last rounds,
        # the small split-K reductions) overlap the other instead of leaving CUs idle.
        main = torch.cuda.current_stream()
        side = _side_stream(dy.device) if _TWO_STREAMS else None
        # LeakyReLU derivative: every block's input (i > 0) is a LeakyReLU output, pooled or not, and a pooled value
        # has the sign of the element the pool routed to — so each dgrad applies the mask of ITS input in its
        # epilogue (fp32, before the one rounding to bf16) and pool backward only routes.  A pool that ends the chain
        # has no dgrad after it and applies the mask itself.
        pooled = None
        unpooled = False     # the dgrad of block i + 1 already carried its gradient through pool i
        for i in range(last, -1, -1):
            if stages[i] == "P":
                if unpooled:
                    unpooled = False
                    continue
                if _FUSE_STEM_POOL_BWD and i == 1 and plans[0] is not None and plans[0].stem:
                    # nothing but the stem's weight gradient reads this pool's input gradient: that kernel builds
                    # its dz tiles from the pooled side itself, and the full-resolution tensor is never written
                    pooled = (gins[i], acts[i] if i == last else None)
                    continue
                g = _pool_bwd_idx(acts[i], gins[i], g, fuse_leaky=(i == last))
                continue
            plan = plans[i]
            pi -= 2
            extra = {"pooled": pooled} if (i == 0 and pooled is not None) else {}
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    g.record_stream(side)
                    gins[i].record_stream(side)
                    for t in (pooled or ()):
                        if t is not None:
                            t.record_stream(side)
                    dw, db = plan.wgrad(gins[i], g, ctx.params[pi], ctx.params[pi + 1], **extra)
                    for t in (dw, db):
                        if t is not None:
                            t.record_stream(main)
            else:
                dw, db = plan.wgrad(gins[i], g, ctx.params[pi], ctx.params[pi + 1], **extra)
            grads[pi], grads[pi + 1] = dw, db
            if i == 0:
                break
            stem_pool = _FUSE_STEM_POOL_BWD and i == 2 and plans[0] is not None and plans[0].stem
            if _FUSE_UNPOOL and stages[i - 1] == "P" and plan.stride == 1 and not stem_pool and acts[i - 1].shape[1] % 64 == 0:
                # a pool in front of this block: its backward is four stores per GEMM row in this dgrad's epilogue
                g = co.conv_dgrad_unpool(g, plan.wt16, acts[i - 1], gins[i - 1], slope=SLOPE)
                unpooled = True
            else:
                g = co.conv_dgrad(g, plan.wt16, acts[i - 1].shape[2:], plan.stride, x_act=acts[i - 1], slope=SLOPE)
        if side is not None:
            main.wait_stream(side)
        ctx.acts = ctx.gins = ctx.plans = ctx.params = None
        return (None, None) + tuple(grads)


class Backbone(nn.Sequential):
    """nn.Sequential whose whole-chain call is one fused autograd node when every child is a Conv2D
    block or a 2x2 max pool (true for ``feat``); otherwise it behaves like nn.Sequential."""

    def _chainable(self):
        for m in self:
            ok = isinstance(m, HipMaxPool2d) or (isinstance(m, nn.Sequential) and len(m) == 2 and
                                                 isinstance(m[0], HipConv2d) and isinstance(m[1], FusedLeakyReLU))
            if not ok:
                return False
        return len(self) > 0 and not isinstance(self[0], HipMaxPool2d)

    def forward(self, x):
        if not (x.is_cuda and self._chainable()):
            return super().forward(x)
        stages, params = [], []
        for m in self:
            if isinstance(m, HipMaxPool2d):
                stages.append("P")
            else:
                stages.append(m[0].stride)
                params += [m[0].weight, m[0].bias]
        return _BackboneFn.apply(x, tuple(stages), *params)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b in bf16 on hipBLASLt (through PyTorch-ROCm: a plain library GEMM, not one of the four
    hand-written kernel sets), fp32 master weight, gradients written into the optimiser's arena."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope=None, p_drop=0.0, seed=0):
        """``slope`` not None: the layer's LeakyReLU(slope) and Dropout(p_drop) run fused behind the GEMM
        (fc_kernels.hip); the GEMM then leaves fp32 (split-K slabs for fc.1) and is rounded to bf16 once."""
        from .train import bf16_shadow
        x16 = x.to(torch.bfloat16)
        w16 = bf16_shadow(weight)
        if w16 is None:
            w16 = weight.detach().to(torch.bfloat16)
        ctx.save_for_backward(x16)
        # not through save_for_backward: every shadow is a view of one flat bf16 arena and shares its version
        # counter, so refreshing ANOTHER parameter's shadow between forward and backward would trip the in-place
        # check although this view's bytes are untouched
        ctx.w16 = w16
        ctx.wb, ctx.in_dtype = (weight, bias), x.dtype
        if x16.dim() != 2 or x16.shape[1] != w16.shape[1]:
            # the reference's own failure mode (a 224x224 input reaches fc.1 with 16384 features),
            # with torch's wording so callers that match on it keep working
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x16.shape[0]}x{x16.shape[-1]} and "
                               f"{w16.shape[1]}x{w16.shape[0]})")
        m, k = x16.shape
        n = w16.shape[0]
        ctx.fused = slope is not None
        split = k >= 8192 and m * n <= (1 << 20) and k % 64 == 0
        if ctx.fused:
            if split:
                S = 8
                part = torch.bmm(x16.view(m, S, k // S).transpose(0, 1), w16.view(n, S, k // S).permute(1, 2, 0),
                                 out_dtype=torch.float32)
            else:
                part = torch.mm(x16, w16.t(), out_dtype=torch.float32)
            y, keep = fo.fc_epilogue_fwd(part, bias, slope, p_drop, seed)
            ctx.epi = (y, keep, slope, p_drop)
            return y
        if split:
            # fc.1 (64 x 4096 x 50176): a plain GEMM has 16 output tiles for 256 CUs and runs at a third of the
            # rate the 411 MB weight stream allows; 8-way split-K as a batched GEMM with fp32 partials fills the
            # chip (0.25 -> 0.09 ms) and is more accurate than bf16 accumulation across the whole K.
            S = 8
            xa = x16.view(m, S, k // S).transpose(0, 1)
            wb = w16.view(n, S, k // S).permute(1, 2, 0)
            y = torch.bmm(xa, wb, out_dtype=torch.float32).sum(0)
            return (y + bias.detach().float()).to(torch.bfloat16)
        return torch.addmm(bias.detach().to(torch.bfloat16), x16, w16.t())

    @staticmethod
    def backward(ctx, dy):
        from .train import write_grad
        (x16,) = ctx.saved_tensors
        w16 = ctx.w16
        weight, bias = ctx.wb
        db_fused = None
        if ctx.fused:      # dropout mask x LeakyReLU derivative and the bias gradient in one pass over dy
            y, keep, slope, p_drop = ctx.epi
            got = {}

            def put_b_fused(sink, acc):
                if sink is None:
                    db = torch.empty(y.shape[1], device=y.device, dtype=torch.float32)
                    got["dz"] = fo.fc_epilogue_bwd(dy, y, keep, slope, p_drop, db=db)
                    return db
                got["dz"] = fo.fc_epilogue_bwd(dy, y, keep, slope, p_drop, db=sink, accumulate=acc)
            db_fused = (write_grad(bias, put_b_fused),)
            dy16 = got["dz"]
            ctx.epi = None
        else:
            dy16 = dy.to(torch.bfloat16).contiguous()
        dx = torch.mm(dy16, w16).to(ctx.in_dtype) if ctx.needs_input_grad[0] else None

        sink_w = getattr(weight, "_yfs_grad_sink", None)
        pending = getattr(weight, "_yfs_factors", None)
        comm = getattr(weight, "_yfs_comm", None)
        multi = comm is not None and getattr(comm, "active", getattr(comm, "world", 1) > 1)
        if pending is not None and sink_w is not None:
            # a second backward before the step: the deferred factors become an ordinary gradient first
            if multi:
                raise L.YfsError("several backward passes per step are not supported together with the deferred Linear "
                                 "gradient under data parallelism (set YFS_DEFER_FC=0)")
            torch.mm(pending[1].t(), pending[0], out_dtype=torch.float32, out=sink_w)
            weight._yfs_factors = None
        elif getattr(weight, "_yfs_defer", False) and sink_w is not None and weight._yfs_grad_fresh:
            # FlatSGD takes this layer's step from the factors of its gradient (train.FlatSGD._step_with_deferred);
            # with several ranks the factors of all ranks are gathered and the mean gradient is 1/world of their product
            if multi:
                xa, da = comm.gather_factors(x16, dy16, owner=weight)
                weight._yfs_factors = (xa, da, 1.0 / comm.world)
            else:
                weight._yfs_factors = (x16, dy16, 1.0)
            weight._yfs_grad_fresh = False
            db = db_fused[0] if db_fused is not None else write_grad(bias, lambda sink, acc: _bias_grad(dy16, sink, acc))
            ctx.wb = ctx.w16 = None
            return dx, None, db, None, None, None

        def put_w(sink, acc):     # bf16 operands, fp32 result straight into the gradient arena (no bf16 round trip)
            if sink is None or acc:
                dw = torch.mm(dy16.t(), x16, out_dtype=torch.float32)
                if sink is None:
                    return dw
                sink.add_(dw)
            else:
                torch.mm(dy16.t(), x16, out_dtype=torch.float32, out=sink)

        def put_b(sink, acc):
            db = dy16.float().sum(0)
            if sink is None:
                return db
            sink.add_(db) if acc else sink.copy_(db)
        dw = write_grad(weight, put_w)
        db = db_fused[0] if db_fused is not None else write_grad(bias, put_b)
        ctx.wb = ctx.w16 = None
        return dx, dw, db, None, None, None


def _bias_grad(dy16, sink, acc):
    db = dy16.float().sum(0)
    if sink is None:
        return db
    sink.add_(db) if acc else sink.copy_(db)


class HipLinear(nn.Linear):
    """nn.Linear with the same parameter names/init; bf16 compute, fp32 master weights.
    ``fused_act`` = slope: the LeakyReLU(slope) that follows the layer (and the Dropout module handed to
    ``fuse_dropout``) run in the layer's own epilogue; the modules themselves stay in the Sequential as identity
    placeholders so the reference's indices and state-dict keys (fc.1, fc.4) are unchanged."""

    def __init__(self, in_features, out_features, fused_act=None):
        super().__init__(in_features, out_features)
        self.weight._yfs_sink_capable = self.bias._yfs_sink_capable = True
        # big enough that skipping the materialised gradient pays (see train.FlatSGD): fc.1 yes, fc.4 no
        self.weight._yfs_defer_capable = (in_features % 128 == 0 and out_features % 128 == 0 and
                                          in_features * out_features >= (1 << 24))
        self.fused_act = fused_act
        self.__dict__["_drop_ref"] = None       # not a submodule: keeps the key names of the reference

    def fuse_dropout(self, drop):
        self.__dict__["_drop_ref"] = drop

    def forward(self, x):
        if not x.is_cuda:
            raise L.YfsError("HipLinear runs on the MI355X only")
        if self.fused_act is None:
            return _LinearFn.apply(x, self.weight, self.bias)
        drop = self.__dict__.get("_drop_ref")
        p = float(drop.p) if (drop is not None and drop.training) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0   # CPU generator: follows torch.manual_seed
        return _LinearFn.apply(x, self.weight, self.bias, float(self.fused_act), p, seed)


class FusedDropout(nn.Dropout):
    """Placeholder at the reference's fc.3: the dropout already ran in fc.1's epilogue (p and train/eval mode are
    read from this module at every call)."""

    def forward(self, x):
        return x


class YoloRegressorModel(nn.Module):
    """Reference YoloRegressorModel (yolo_toy_regressor_torch.py:213-269).

    ``head_out`` = 4 reproduces the reference ([xc,yc,w,h]); 7*7*30 = 1470 gives the YOLOv1 S=7,B=2,C=20
    head that BASELINE.json's train-step config uses (build-defined: the reference stops at 4 outputs).
    ``image_size`` sizes fc.1 the way the reference's 448 dummy forward does (:255-257).
    """

    def __init__(self, head_out=4, image_size=448, p_drop=0.2):
        super().__init__()
        pool = HipMaxPool2d
        self.feat = Backbone(
            Conv2D(3, 192, 7, s=2), pool(),
            Conv2D(192, 256, 3), pool(),
            Conv2D(256, 128, 1), Conv2D(128, 256, 3), Conv2D(256, 256, 1), Conv2D(256, 512, 3), pool(),
            Conv2D(512, 256, 1), Conv2D(256, 512, 3), Conv2D(512, 256, 1), Conv2D(256, 512, 3),
            Conv2D(512, 256, 1), Conv2D(256, 512, 3), Conv2D(512, 256, 1), Conv2D(256, 512, 3),
            Conv2D(512, 512, 1), Conv2D(512, 1024, 3), pool(),
            Conv2D(1024, 512, 1), Conv2D(512, 1024, 3), Conv2D(1024, 512, 1), Conv2D(512, 1024, 3),
            Conv2D(1024, 1024, 3),
            Conv2D(1024, 1024, 3, s=2),
            Conv2D(1024, 1024, 3), Conv2D(1024, 1024, 3),
        )
        hw = image_size
        for m in self.feat:
            hw = hw // 2 if isinstance(m, HipMaxPool2d) else co.out_hw(hw, hw, m[0].kernel_size, m[0].stride)[0]
        flat = 1024 * hw * hw
        self.fc = nn.Sequential(
            nn.Flatten(),
            HipLinear(flat, 4096, fused_act=SLOPE if _FUSE_FC else None),
            FusedLeakyReLU(0.1) if _FUSE_FC else nn.LeakyReLU(0.1, inplace=True),
            FusedDropout(p_drop) if _FUSE_FC else nn.Dropout(p_drop),
            HipLinear(4096, head_out),
        )
        if _FUSE_FC:
            self.fc[1].fuse_dropout(self.fc[3])

    def forward(self, x):
        x = self.feat(x)
        # the reference flattens NCHW; x is NHWC in memory, so flatten() permutes (6 MB at batch 64)
        return self.fc(x).float()
