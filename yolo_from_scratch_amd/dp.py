"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device (yolo_toy_regressor_torch.py:375); this is new work.
Design for the MI355X node:
  * gradients already live in ONE flat fp32 arena (train.FlatSGD), so buckets are plain slices of it —
    no gather/scatter copies;
  * buckets are cut in REVERSE parameter order: backward produces fc.4, fc.1 first and fc.1 alone is 77 %
    of the payload, so the biggest transfer starts first and hides under the whole conv backward;
  * each bucket's all-reduce is issued on a side stream as soon as every parameter in it has reported its
    gradient (``notify``), ordered after the producing kernels by an event; ``finish`` makes the compute
    stream wait for all of them before clip_grad_norm_/SGD read the arena (the reference clips AFTER
    backward, :331-333, so the reduced gradient is what must be clipped);
  * xGMI is point-to-point (7 links per GPU): large messages, but the LAST bucket (the first layers' gradients,
    produced at the very end of backward) has nothing left to hide under, so it must be small.  fc.1 (820 MB) is
    one message and goes first; the 242 MB of conv gradients travel as ~32 MB buckets while the remaining
    dgrad/wgrad kernels run; the gradients of the first ``tail_bytes`` of the arena (feat.0-6, a few MB) are cut
    at ``tail_bucket_bytes`` so the exposed tail is a latency-sized message (a single 256 MB conv bucket would
    leave ~1.4 ms of a ~14.5 ms step exposed on 8 GPUs).
Averaging: SUM all-reduce, then the 1/world factor is folded into the bf16->fp32 copy-back (compressed
mode) or applied as one in-place scale on the side stream (fp32 mode).
"""
import torch
import torch.distributed as dist


_PRODUCERS = []


def register_producer_stream(stream):
    """Streams other than the current one on which gradient kernels are enqueued (model.py runs the weight
    gradients on a side stream); a bucket's all-reduce waits for all of them."""
    if stream not in _PRODUCERS:
        _PRODUCERS.append(stream)


class GradComm:
    def __init__(self, world_size=None, bucket_bytes=32 << 20, compress_bf16=False, group=None,
                 tail_bytes=16 << 20, tail_bucket_bytes=4 << 20, force=False):
        """``force``: run the whole exchange (buckets, side stream, collectives, factor gather) even in a group of ONE
        rank, where it must change nothing — the only way to execute the RCCL code path on a one-GPU box
        (tests/test_zy_rccl_one_rank.py)."""
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.active = self.world > 1 or force
        self.bucket_bytes = bucket_bytes
        self.tail_bytes, self.tail_bucket_bytes = tail_bytes, min(tail_bucket_bytes, bucket_bytes)
        self.compress = compress_bf16
        self.buckets = []          # (start, end) element ranges of the flat arena, in issue order
        self.param_bucket = {}     # id(param) -> bucket index
        self.pending = []          # params not yet reported, per bucket
        self.flat = None
        self.stream = None
        self._issued = []
        self._c16 = None
        self._host_waits = []      # (work, chunk) of collectives whose completion the host must wait for (gloo)
        self._deferred = []        # gloo on CUDA tensors: buckets whose collective runs in reduce()

    # -- layout -----------------------------------------------------------------------------------
    def attach(self, params, offsets, flat_g, skip_ids=()):
        """params / offsets as laid out by FlatSGD (registration order); buckets are cut from the END.
        ``skip_ids``: id()s of parameters whose gradient is never materialised (FlatSGD's deferred Linear weights:
        their factors travel through ``gather_factors`` instead); no bucket covers their slice of the arena."""
        self.flat = flat_g
        per, per_tail, tail = max(1, self.bucket_bytes // 4), max(1, self.tail_bucket_bytes // 4), self.tail_bytes // 4
        self.buckets, self.param_bucket = [], {}
        cur_hi, cur_lo, members = None, None, []
        for p, (off, n) in zip(reversed(params), reversed(offsets)):
            if id(p) in skip_ids:
                if members:
                    self._close(cur_lo, cur_hi, members)
                    cur_hi, members = None, []
                continue
            if cur_hi is None:
                cur_hi = off + n
            cur_lo = off
            members.append(p)
            if cur_hi - cur_lo >= (per_tail if cur_hi <= tail else per):
                self._close(cur_lo, cur_hi, members)
                cur_hi, members = None, []
        if members:
            self._close(cur_lo, cur_hi, members)
        if flat_g.is_cuda:
            self.stream = torch.cuda.Stream(device=flat_g.device)
            if self.compress:
                self._c16 = torch.empty(max(e - s for s, e in self.buckets), device=flat_g.device, dtype=torch.bfloat16)
        self.begin_step()

    def _close(self, lo, hi, members):
        idx = len(self.buckets)
        self.buckets.append((lo, hi))
        for p in members:
            self.param_bucket[id(p)] = idx

    # -- per step ------------------------------------------------------------------------------------
    def begin_step(self):
        self.pending = [0] * len(self.buckets)
        for idx in self.param_bucket.values():
            self.pending[idx] += 1
        self._issued = [False] * len(self.buckets)

    def notify(self, param):
        """Called by the autograd nodes right after they enqueue the kernel that writes param's gradient."""
        if not self.active:
            return
        idx = self.param_bucket.get(id(param))
        if idx is None or self._issued[idx]:
            return
        self.pending[idx] -= 1
        if self.pending[idx] == 0:
            self._issue(idx)

    def _issue(self, idx):
        """Start bucket idx's all-reduce.  Called from the autograd thread in the middle of backward, so it must never
        wait on the HOST for a peer: a rank that blocked here could not enqueue the rest of its backward, and on a
        backend whose collectives are host-synchronous (gloo) two ranks would serialise on every bucket (round 1's
        one-GPU gloo rehearsal ended in exactly such a wait, gpurun_out/dp2.log).  Every collective is therefore
        started with async_op=True:
          * nccl (= RCCL): the collective is enqueued on RCCL's own stream behind the exchange stream; ``work.wait()``
            only makes the exchange stream wait for it (a stream dependency, no host block), and the 1/world scale is
            queued behind it;
          * gloo on CPU tensors (tests): the Work object is kept and ``reduce`` — called by FlatSGD.step after backward
            has been enqueued completely — waits for it and applies the scale;
          * gloo on CUDA tensors (one-GPU rehearsals): the bucket is only marked; ``reduce`` runs the collectives.
        Order: a bucket is issued when its last parameter reports, and ``reduce`` issues the rest in index order; both are
        functions of the autograd graph alone, which is identical on every rank, so all ranks issue the same sequence."""
        self._issued[idx] = True
        lo, hi = self.buckets[idx]
        chunk = self.flat[lo:hi]
        host_sync = dist.is_initialized() and dist.get_backend(self.group) != "nccl"
        if self.stream is None:                               # CPU tensors (gloo tests)
            work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._host_waits.append((work, chunk, None))
            return
        if host_sync:
            # gloo on CUDA tensors (one-GPU rehearsals): the backend reads the tensor from the host side, so the producing
            # kernels must have finished first.  Nothing of that may happen inside backward - not even a wait on this
            # rank's own GPU work - so the bucket is only marked here and ``reduce`` runs the collective.
            self._deferred.append(idx)
            return
        cur = torch.cuda.current_stream()
        evs = []
        for st in [cur] + [p for p in _PRODUCERS if p != cur and p.device == cur.device]:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        with torch.cuda.stream(self.stream):
            for ev in evs:
                self.stream.wait_event(ev)
            buf = chunk
            if self.compress:
                # one staging buffer for all buckets: safe because every use is ordered on the exchange stream
                buf = self._c16[: hi - lo]
                buf.copy_(chunk)
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if work is not None:
                work.wait()                                   # nccl: stream-level wait on the exchange stream
            if self.compress:
                torch.mul(buf, 1.0 / self.world, out=chunk)
            else:
                chunk.mul_(1.0 / self.world)

    def gather_factors(self, x16, dy16):
        """All-gather of the two factors of a Linear weight gradient (dW = dy^T x): [B, In] and [B, Out] per rank become
        [world*B, ...].  For fc.1 that is 6.9 MB per rank on the wire instead of an 807 MB all-reduce, and the averaged
        gradient sum_r dy_r^T x_r / world is formed inside the fused wgrad + SGD kernel, never in memory.  Runs on the
        exchange stream; ``reduce`` (called by FlatSGD.step) makes the compute stream wait for it."""
        if not self.active:
            return x16, dy16
        outs = []
        if self.stream is None:                               # CPU / gloo (tests)
            for t in (x16, dy16):
                out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                dist.all_gather(list(out.chunk(self.world)), t.contiguous(), group=self.group)
                outs.append(out)
            return tuple(outs)
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            for t in (x16, dy16):
                t = t.contiguous()
                t.record_stream(self.stream)
                out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                dist.all_gather(list(out.chunk(self.world)), t, group=self.group)
                out.record_stream(cur)
                outs.append(out)
        return tuple(outs)

    def reduce(self, flat_g):
        """Issue whatever has not been issued yet (parameters that never reported) and wait for everything."""
        if self.active:
            for idx in range(len(self.buckets)):
                if not self._issued[idx]:
                    self._issue(idx)
            if self._deferred:                                # gloo on CUDA tensors: backward is fully enqueued by now
                cur = torch.cuda.current_stream()
                self.stream.wait_stream(cur)
                for p in _PRODUCERS:
                    if p.device == cur.device:
                        self.stream.wait_stream(p)
                self.stream.synchronize()                     # this rank's own gradients are complete
                with torch.cuda.stream(self.stream):
                    works = []
                    for idx in sorted(self._deferred):        # the same order on every rank
                        lo, hi = self.buckets[idx]
                        chunk = flat_g[lo:hi]
                        buf = chunk.to(torch.bfloat16) if self.compress else chunk
                        works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk, buf))
                    for work, chunk, buf in works:
                        # waited with the exchange stream current: the backend orders its copy back to the device before
                        # the stream that is current at wait(), and the scale below must come after that copy
                        work.wait()
                        if self.compress:
                            torch.mul(buf, 1.0 / self.world, out=chunk)
                        else:
                            chunk.mul_(1.0 / self.world)
                self._deferred = []
            for work, chunk, staged in self._host_waits:      # gloo on CPU tensors
                if work is not None:
                    work.wait()
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        if staged is not None:
                            torch.mul(staged, 1.0 / self.world, out=chunk)
                        else:
                            chunk.mul_(1.0 / self.world)
                else:
                    chunk.mul_(1.0 / self.world)
            self._host_waits = []
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        self.begin_step()


def broadcast_parameters(flat_p, src=0, group=None):
    """Identical initial weights on every rank.  ``flat_p`` is the optimiser's flat
    fp32 arena, or the FlatSGD itself (then its bf16 shadow is refreshed as well)."""
    opt = None
    if hasattr(flat_p, "flat_p"):
        opt, flat_p = flat_p, flat_p.flat_p
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_p, src=src, group=group)
        if opt is not None:
            opt.sync_shadow()
