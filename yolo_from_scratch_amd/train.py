"""Step driver — the reference's train/eval surface (yolo_toy_regressor_torch.py:272-339) plus the
MI355X-native optimiser: all parameters, gradients and momentum buffers live in three flat fp32 arenas
(1.06 GB each for the 266 M-parameter model; HBM is 288 GB), backward writes weight gradients straight
into the gradient arena, and clip_grad_norm_ + SGD run as two HBM-bound launches over the arenas.
"""
import gc
import os

import torch
import torch.nn as nn

from . import _lib as L
from .boxes import iou_xywh_torch


# ---- yolo_toy_regressor_torch.py:272-280 --------------------------------------------------------
def yolo_v1_lr_schedule(epoch, total_epochs=135, warmup=5):
    if epoch < warmup:
        return 1e-3 + (1e-2 - 1e-3) * (epoch + 1) / max(1, warmup)
    elif epoch < warmup + 75:
        return 1e-2
    elif epoch < warmup + 75 + 30:
        return 1e-3
    else:
        return 1e-4


# ---- yolo_toy_regressor_torch.py:282-296 --------------------------------------------------------
class EarlyStoppingMax:
    def __init__(self, patience=12, min_delta=0.0):
        self.best = -float("inf")
        self.patience = patience
        self.min_delta = min_delta
        self.count = 0

    def step(self, metric):
        improved = metric > self.best + self.min_delta
        if improved:
            self.best = metric
            self.count = 0
        else:
            self.count += 1
        return improved, self.count >= self.patience


class FlatSGD(torch.optim.Optimizer):
    """SGD(momentum, weight_decay) with fused clip_grad_norm_ over flat arenas.

    Construction re-homes every parameter into one flat fp32 buffer (values preserved, layouts
    preserved) and gives each a persistent ``.grad`` view into a flat gradient buffer.  The conv and
    linear autograd nodes of this package write gradients directly into those views (``_yfs_grad_sink``);
    anything else reaches them through autograd's normal in-place accumulation.
    ``comm`` (optional) is a gradient-exchange object with ``reduce(flat_grads)`` (see dp.py).
    """

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=None, comm=None, overlap_tail=False):
        params = list(params)
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        if not params or not all(p.is_cuda for p in params):
            raise L.YfsError("FlatSGD needs CUDA (MI355X) parameters")
        dev = params[0].device
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]          # 256-byte aligned slices
        total = sum(sizes)
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_p16 = torch.zeros(total, device=dev, dtype=torch.bfloat16)   # bf16 shadow, rewritten by every step
        self.offsets = []
        off = 0
        for p, sz in zip(params, sizes):
            n = p.numel()
            if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                raise L.YfsError("FlatSGD: parameters must be dense")
            pv = torch.as_strided(self.flat_p, p.shape, p.stride(), off)
            pv.copy_(p.data)
            p.data = pv
            gv = torch.as_strided(self.flat_g, p.shape, p.stride(), off)
            p.grad = gv
            p._yfs_grad_sink = gv
            p._yfs_grad_fresh = bool(getattr(p, "_yfs_sink_capable", False))
            p._yfs_bf16 = torch.as_strided(self.flat_p16, p.shape, p.stride(), off)
            self.offsets.append((off, n))
            off += sz
        self.total = total
        self.sync_shadow()
        self.max_norm = max_norm
        self.comm = comm
        self._params = params
        # Large Linear weights (fc.1: 205 M of the 266 M parameters) take their step straight from the factors of their
        # gradient (dy^T x) in one fused kernel — the 822 MB gradient is never written, summed or re-read, and under
        # data parallelism the factors are all-gathered (6.9 MB per rank) instead of all-reducing 807 MB (dp.GradComm.
        # gather_factors).  ``p.grad`` of such a parameter is then NOT updated by backward (YFS_DEFER_FC=0 restores the
        # materialised gradient).
        self._defer = (comm is None or hasattr(comm, "gather_factors")) and os.environ.get("YFS_DEFER_FC", "1") != "0"
        skip = set()
        for p, (o, n), sz in zip(params, self.offsets, sizes):
            p._yfs_region = (o, sz)
            p._yfs_defer = bool(self._defer and getattr(p, "_yfs_defer_capable", False) and n == sz)
            if p._yfs_defer:
                skip.add(id(p))
        if comm is not None:
            if skip:
                comm.attach(params, self.offsets, self.flat_g, skip_ids=skip)
            else:
                comm.attach(params, self.offsets, self.flat_g)
            for p in params:
                p._yfs_comm = comm
        self._first = True
        # overlap_tail: the fused step of a deferred Linear weight (fc.1: 3.7 GB of HBM traffic, no matrix work) is
        # enqueued on the weight-gradient stream and step() returns with it in flight, so it runs under the next
        # forward pass's convolutions (matrix-bound, little HBM traffic).  Everything in this package that reads such a
        # parameter waits for it first (bf16_shadow -> the Linear forward, step(), train_one_epoch's end); code that reads
        # the parameter tensor itself right after step() must call join().  Off by default for that reason;
        # train_one_epoch switches it on for its own loop.
        self.overlap_tail = bool(overlap_tail)
        self._tail = None
        self._tail_recorded = False
        for p in params:
            if p._yfs_defer:
                p._yfs_tail_owner = self
        self._sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        ws_bytes = L.lib().yfs_sumsq_workspace_bytes(L.i64(total))
        self._ws = torch.empty(ws_bytes // 4 + 1, device=dev, dtype=torch.float32)
        self._ws_bytes = ws_bytes

    def zero_grad(self, set_to_none=True):
        """Gradients stay allocated (they are arena views).  Parameters owned by this package's layers are
        marked fresh: their next backward overwrites the view instead of adding.  Any other parameter
        receives its gradient through autograd's in-place accumulation, so its view is cleared here."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not p._yfs_grad_sink:
                    p.grad = p._yfs_grad_sink
                if getattr(p, "_yfs_sink_capable", False):
                    p._yfs_grad_fresh = True
                else:
                    p._yfs_grad_sink.zero_()
                    p._yfs_grad_fresh = False

    def _finish_grads(self):
        """A sink-capable parameter that took no part in this step's backward still holds last step's
        gradient: clear it (autograd would have left .grad = None)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p._yfs_grad_fresh:
                    p._yfs_grad_sink.zero_()
                    p._yfs_grad_fresh = False
                    grad_ready(p)
                elif not getattr(p, "_yfs_sink_capable", False):
                    grad_ready(p)       # gradient came through plain autograd: complete by now

    @torch.no_grad()
    def sync_shadow(self):
        """Recompute the bf16 shadow from the fp32 master weights.  Done at construction; in-place writes to a
        parameter (load_state_dict, p.copy_) are noticed through the tensor version and refreshed lazily by
        bf16_shadow(); call this after writing through ``p.data`` / the flat arena, which no version tracks."""
        self.flat_p16.copy_(self.flat_p)
        for p in self._all_params():
            p._yfs_bf16_ver = p._version

    def _all_params(self):
        for group in self.param_groups:
            yield from group["params"]

    def join(self):
        """The current stream waits for a Linear step that ``overlap_tail`` left running on the side stream."""
        if self._tail is not None:
            torch.cuda.current_stream().wait_event(self._tail)

    @torch.no_grad()
    def grad_norm(self):
        L.check(L.lib().yfs_sumsq_f32(L.ptr(self.flat_g), L.i64(self.total), L.ptr(self._sumsq), L.ptr(self._ws),
                                      L.c_size(self._ws_bytes), L.stream()), "yfs_sumsq_f32")
        return self._sumsq

    def _sumsq_range(self, lo, hi, out):
        n = hi - lo
        L.check(L.lib().yfs_sumsq_f32(L.ptr(self.flat_g[lo:hi]), L.i64(n), L.ptr(out), L.ptr(self._ws), L.c_size(self._ws_bytes),
                                      L.stream()), "yfs_sumsq_f32")

    def _step_with_deferred(self, deferred, max_norm, need_norm=False):
        """Parameters in ``deferred`` carry (x16, dy16) instead of a gradient.  Norm: materialised ranges by the
        sumsq kernel, deferred ones as ||dy^T x||_F^2 = sum((dy dy^T) * (x x^T)); then the plain SGD kernel over the
        materialised ranges and the fused wgrad + SGD kernel per deferred parameter."""
        from . import fc_ops as fo
        g = self.param_groups[0]
        regions = sorted(p._yfs_region for p in deferred)
        ranges, lo = [], 0
        for o, sz in regions:
            if o > lo:
                ranges.append((lo, o))
            lo = o + sz
        if lo < self.total:
            ranges.append((lo, self.total))
        sumsq = None
        if need_norm or (max_norm is not None and max_norm > 0):
            parts = torch.zeros(len(ranges) + len(deferred), device=self.flat_g.device, dtype=torch.float32)
            for i, (a, b) in enumerate(ranges):
                self._sumsq_range(a, b, parts[i:i + 1])
            for j, p in enumerate(deferred):
                x16, dy16, gs = p._yfs_factors
                gx = torch.mm(x16, x16.t(), out_dtype=torch.float32)
                gy = torch.mm(dy16, dy16.t(), out_dtype=torch.float32)
                parts[len(ranges) + j] = (gx * gy).sum() * (gs * gs)
            torch.sum(parts, dim=0, keepdim=True, out=self._sumsq)
            sumsq = self._sumsq
            if need_norm and not bool(torch.isfinite(self._sumsq).item()):     # GradScaler semantics: skip the step
                for p in deferred:
                    p._yfs_factors = None
                return False
            if not (max_norm is not None and max_norm > 0):
                sumsq = None
        first = L.c_int(1 if self._first else 0)
        for a, b in ranges:
            L.check(L.lib().yfs_sgd_step_f32(L.ptr(self.flat_p[a:b]), L.ptr(self.flat_g[a:b]), L.ptr(self.flat_m[a:b]), L.i64(b - a),
                                             L.ptr(sumsq), L.c_float(max_norm or 0.0), L.c_float(g["lr"]),
                                             L.c_float(g["momentum"]), L.c_float(g["weight_decay"]), first,
                                             L.ptr(self.flat_p16[a:b]), L.stream()), "yfs_sgd_step_f32")
        def linear_steps():
            for p in deferred:
                x16, dy16, gs = p._yfs_factors
                o, sz = p._yfs_region
                shape = tuple(p.shape)
                fo.linear_wgrad_sgd(x16, dy16, self.flat_p[o:o + sz].view(shape), self.flat_m[o:o + sz].view(shape),
                                    self.flat_p16[o:o + sz].view(shape), sumsq, max_norm, g["lr"], g["momentum"],
                                    g["weight_decay"], self._first, gs)
                p._yfs_factors = None

        if self.overlap_tail:
            from .model import _side_stream
            side = _side_stream(self.flat_p.device)
            side.wait_stream(torch.cuda.current_stream())      # the norm and the factors are produced on this stream
            with torch.cuda.stream(side):
                for p in deferred:
                    for t in p._yfs_factors[:2]:
                        t.record_stream(side)
                if not self._tail_recorded:      # the arenas must outlive that stream's work if the optimiser is dropped
                    for t in (self.flat_p, self.flat_m, self.flat_p16, self._sumsq):
                        t.record_stream(side)
                    self._tail_recorded = True
                linear_steps()
                self._tail = torch.cuda.Event()
                self._tail.record(side)
        else:
            linear_steps()
        self._first = False
        return True

    @torch.no_grad()
    def step(self, closure=None, max_norm="default", grad_scale=None):
        """``grad_scale`` (GradScaler path, train_one_epoch): every gradient of this step carries the loss scale
        1/grad_scale; it is removed here — from the materialised gradients in place and from the factors of a deferred
        Linear weight through the fused kernel's own scale — AFTER the data-parallel exchange and BEFORE the norm, so
        clipping sees the true averaged gradient.  Returns False (and leaves the parameters untouched) when that
        gradient is not finite, which is what torch.amp.GradScaler.step does; True otherwise."""
        if max_norm == "default":
            max_norm = self.max_norm
        self.join()              # last step's overlapped Linear update reads the norm buffer this step rewrites
        self._tail = None
        self._finish_grads()
        if self.comm is not None:
            self.comm.reduce(self.flat_g)
        deferred = [p for p in self._params if getattr(p, "_yfs_factors", None) is not None]
        check = grad_scale is not None
        if check and grad_scale != 1.0:
            self.flat_g.mul_(grad_scale)
            for p in deferred:
                x16, dy16, gs = p._yfs_factors
                p._yfs_factors = (x16, dy16, gs * grad_scale)
        if deferred:
            return self._step_with_deferred(deferred, max_norm, need_norm=check)
        g = self.param_groups[0]
        sumsq = None
        if check or (max_norm is not None and max_norm > 0):
            sumsq = self.grad_norm()
            if check and not bool(torch.isfinite(sumsq).item()):
                return False
            if not (max_norm is not None and max_norm > 0):
                sumsq = None
        L.check(L.lib().yfs_sgd_step_f32(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.flat_m), L.i64(self.total),
                                         L.ptr(sumsq), L.c_float(max_norm or 0.0), L.c_float(g["lr"]),
                                         L.c_float(g["momentum"]), L.c_float(g["weight_decay"]),
                                         L.c_int(1 if self._first else 0), L.ptr(self.flat_p16), L.stream()),
                "yfs_sgd_step_f32")
        self._first = False
        return True


def bf16_shadow(param):
    """bf16 copy of ``param`` kept current by FlatSGD.step (None when the parameter is not in a FlatSGD)."""
    sh = getattr(param, "_yfs_bf16", None)
    if sh is None:
        return None
    owner = getattr(param, "_yfs_tail_owner", None)
    if owner is not None:
        owner.join()
    if param._version != param._yfs_bf16_ver:      # written in place since the last step (e.g. load_state_dict)
        with torch.no_grad():
            sh.copy_(param.detach())
        param._yfs_bf16_ver = param._version
    return sh


def write_grad(param, value_fn):
    """Used by this package's autograd nodes: if ``param`` has a gradient sink, let ``value_fn(sink, accumulate)``
    write into it and return None (autograd then has nothing to accumulate); otherwise return the gradient."""
    sink = getattr(param, "_yfs_grad_sink", None)
    if sink is None:
        return value_fn(None, False)
    value_fn(sink, not param._yfs_grad_fresh)
    param._yfs_grad_fresh = False
    grad_ready(param)
    return None


def grad_ready(param):
    """Tell the gradient exchange (if any) that param's gradient kernel has been enqueued."""
    comm = getattr(param, "_yfs_comm", None)
    if comm is not None:
        comm.notify(param)


# ---- yolo_toy_regressor_torch.py:298-312 --------------------------------------------------------
@torch.no_grad()
def evaluate(model, loader, device, loss_fn):
    model.eval()
    total_loss = torch.zeros((), dtype=torch.float64, device=device)   # summed on the device: see train_one_epoch
    total_iou = 0.0
    n = 0
    for xb, yb in loader:
        xb = xb.to(device, non_blocking=True)
        yb = yb.to(device, non_blocking=True)
        preds = model(xb)
        loss = loss_fn(preds, yb)
        total_loss += loss.detach().double() * xb.size(0)
        if preds.dim() == 2 and preds.shape[1] == 4:
            total_iou += iou_xywh_torch(preds, yb) * xb.size(0)
        n += xb.size(0)
    return total_loss.item() / n, total_iou / n


# ---- yolo_toy_regressor_torch.py:314-339 --------------------------------------------------------
def train_one_epoch(model, loader, device, opt, scaler, loss_fn, max_norm=1.0):
    """Same call and return as the reference.  The model computes in bf16 internally, so ``scaler`` may be a
    disabled GradScaler (or None).  An enabled one works too: with FlatSGD the loss scale is removed inside
    ``opt.step`` (after the data-parallel exchange, before the norm and the clip, deferred Linear factors included),
    a non-finite gradient skips the step and halves the scale as GradScaler does; the scale is never grown (bf16 has
    fp32's exponent range, growth buys nothing).  With any other optimiser the reference's own sequence runs."""
    model.train()
    # The reference adds loss.item() every batch (:337), which stalls the host on the stream each step; the same
    # sum is kept on the device in fp64 (what the Python float sum is) and read once at the end of the epoch.
    running = torch.zeros((), dtype=torch.float64, device=device)
    n = 0
    flat = isinstance(opt, FlatSGD)
    # A generation-2 garbage collection walks every object torch created at import (~75 ms) while the GPU idles;
    # freezing what exists now keeps full collections during the epoch to the few objects the loop itself creates.
    # The caller's heap is thawed again before returning.
    gc.collect()
    gc.freeze()
    # The loss scale is followed on the host (read from the scaler once per epoch; this loop is the only writer), so the
    # scaler path pays ONE host read per step, opt.step's finite check — what torch.amp.GradScaler.step itself pays.
    scale = float(scaler.get_scale()) if (flat and scaler is not None and scaler.is_enabled()) else 1.0
    # Inside this loop nothing reads fc.1's weight between a step and the next forward pass, so the fused step of that
    # weight runs behind the next batch's convolutions (FlatSGD.overlap_tail); joined before returning.
    overlap_before = opt.overlap_tail if flat else False
    if flat:
        opt.overlap_tail = True
    try:
        for xb, yb in loader:
            xb = xb.to(device, non_blocking=True)
            yb = yb.to(device, non_blocking=True)
            opt.zero_grad(set_to_none=True)
            preds = model(xb)
            loss = loss_fn(preds, yb)
            if scaler is not None and scaler.is_enabled():
                scaler.scale(loss).backward()
                if flat:
                    ok = opt.step(max_norm=max_norm, grad_scale=1.0 / scale)
                    if not ok:
                        scale *= scaler.get_backoff_factor()
                    scaler.update(scale)
                else:
                    scaler.unscale_(opt)
                    if max_norm is not None:
                        nn.utils.clip_grad_norm_(model.parameters(), max_norm)
                    scaler.step(opt)
                    scaler.update()
            else:
                loss.backward()
                if flat:
                    opt.step(max_norm=max_norm)
                else:
                    if max_norm is not None:
                        nn.utils.clip_grad_norm_(model.parameters(), max_norm)
                    opt.step()
            running += loss.detach().double() * xb.size(0)
            n += xb.size(0)
    finally:
        gc.unfreeze()
        if flat:
            opt.overlap_tail = overlap_before
            opt.join()
    train_one_epoch.last_count = n
    return running.item() / n
