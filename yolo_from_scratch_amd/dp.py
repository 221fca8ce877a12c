"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device (yolo_toy_regressor_torch.py:375); this is new work.
Design for the MI355X node:
  * gradients already live in ONE flat fp32 arena (train.FlatSGD), so buckets are plain slices of it —
    no gather/scatter copies;
  * buckets are cut in REVERSE parameter order: backward produces fc.4, fc.1 first and fc.1 alone is 77 %
    of the payload, so the biggest transfer starts first and hides under the whole conv backward;
  * each bucket's all-reduce is issued as soon as every parameter in it has reported its gradient (``notify``),
    from the stream that wrote the last of them (RCCL orders the collective after that stream's kernels on its own
    stream); ``reduce`` makes the compute stream wait for all of them before clip_grad_norm_/SGD read the arena
    (the reference clips AFTER backward, :331-333, so the reduced gradient is what must be clipped);
  * xGMI is point-to-point (7 links per GPU): large messages, but the LAST bucket (the first layers' gradients,
    produced at the very end of backward) has nothing left to hide under, so it must be small.  fc.1 (820 MB) is
    one message and goes first; the 242 MB of conv gradients travel as ~32 MB buckets while the remaining
    dgrad/wgrad kernels run; the gradients of the first ``tail_bytes`` of the arena (feat.0-6, a few MB) are cut
    at ``tail_bucket_bytes`` so the exposed tail is a latency-sized message (a single 256 MB conv bucket would
    leave ~1.4 ms of a ~14.5 ms step exposed on 8 GPUs).
Averaging: SUM all-reduce, then the 1/world factor is folded into the bf16->fp32 copy-back (compressed
mode) or applied in ``reduce`` as one in-place scale per contiguous run of buckets (fp32 mode).
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
    """Owns no stream.  Round 2 ran every collective behind an exchange stream of its own; with the weight-gradient
    stream and the backend's stream that made four or more streams per rank on a runtime that multiplexes streams onto
    GPU_MAX_HW_QUEUES (4 on the MI355X boxes) in-order hardware queues, where a wait packet of one stream holds up every
    other stream's packets behind it in the same queue.  The collective only needs to be ORDERED
    after the kernels that wrote the bucket, and RCCL already runs it on its own stream behind an event of the stream
    that is current at the call; so it is issued from whichever stream wrote the bucket's last gradient, nothing waits
    for it inside backward, and ``reduce`` makes the compute stream wait for all of them and applies the 1/world
    factor.  Streams per rank under RCCL: compute, weight gradients, RCCL's own."""

    def __init__(self, world_size=None, bucket_bytes=32 << 20, compress_bf16=False, group=None,
                 tail_bytes=16 << 20, tail_bucket_bytes=4 << 20, force=False):
        """``force``: run the whole exchange (buckets, collectives, factor gather) even in a group of ONE rank, where it
        must change nothing — the only way to execute the RCCL code path on a one-GPU box
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
        self.exchange_streams = 0  # streams this object created (none: see the class docstring)
        self._issued = []
        self._c16 = {}             # bucket index -> bf16 staging buffer (compressed mode)
        self._works = []           # (work or None, bucket index or None, staging buffer or None), in issue order
        self._deferred = []        # host-waited backend (gloo) on CUDA tensors: buckets whose collective runs in reduce()
        self._gathers = []         # ... and the factor gathers: (owner parameter, x16, dy16)
        self._scale_ranges = []    # merged element ranges covered by buckets (fp32 mode: one scale launch per range)

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
        # neighbouring buckets are separated only by the arena's alignment padding (< 64 zero elements): one scale
        # launch per run of neighbours instead of one per bucket
        self._scale_ranges = []
        for lo, hi in sorted(self.buckets):
            if self._scale_ranges and lo - self._scale_ranges[-1][1] < 64:
                self._scale_ranges[-1][1] = hi
            else:
                self._scale_ranges.append([lo, hi])
        self._c16 = {}
        if flat_g.is_cuda and self.compress:
            for i, (s, e) in enumerate(self.buckets):      # one staging buffer per bucket: 0.5 B per gradient element
                self._c16[i] = torch.empty(e - s, device=flat_g.device, dtype=torch.bfloat16)
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

    def _host_waited(self):
        """True for a backend whose collectives the HOST must wait for (gloo); False for RCCL, whose collectives are
        stream-ordered, and for the patched-in fakes of the one-GPU tests (no process group)."""
        return dist.is_initialized() and dist.get_backend(self.group) != "nccl"

    @staticmethod
    def _order_after_producers(cur):
        """Make ``cur`` wait for what the other gradient-producing streams have enqueued so far."""
        for st in _PRODUCERS:
            if st != cur and st.device == cur.device:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)

    def _issue(self, idx):
        """Start bucket idx's all-reduce.  Called from the autograd thread in the middle of backward, so it must never
        wait on the HOST for a peer (a rank that blocked here could not enqueue the rest of its backward; round 1's
        one-GPU gloo rehearsal ended in exactly such a wait, gpurun_out/dp2.log) and it must not put a wait for the
        collective in front of later backward kernels either.  Every collective is started with async_op=True:
          * nccl (= RCCL): the backend's stream waits for an event of the CURRENT stream (the one whose kernel wrote
            the bucket's last gradient: the weight-gradient stream for conv buckets) and runs the collective; the Work
            object is kept, ``reduce`` makes the compute stream wait for it.  When the call comes from the compute
            stream (Linear layers at the start of backward, leftovers at the end) that stream first waits for the
            weight-gradient stream's kernels enqueued so far;
          * gloo on CPU tensors (tests): the Work object is kept and ``reduce`` waits for it;
          * gloo on CUDA tensors (one-GPU rehearsals): the bucket is only marked; ``reduce`` runs the collectives.
        Order: a bucket is issued when its last parameter reports, and ``reduce`` issues the rest in index order; both are
        functions of the autograd graph alone, which is identical on every rank, so all ranks issue the same sequence."""
        self._issued[idx] = True
        lo, hi = self.buckets[idx]
        chunk = self.flat[lo:hi]
        if not chunk.is_cuda:                                 # CPU tensors (gloo tests)
            self._works.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), idx, None))
            return
        if self._host_waited():
            self._deferred.append(idx)
            return
        cur = torch.cuda.current_stream()
        self._order_after_producers(cur)
        buf = None
        if self.compress:
            buf = self._c16[idx]
            buf.copy_(chunk)
        work = dist.all_reduce(buf if buf is not None else chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, idx, buf))

    def gather_factors(self, x16, dy16, owner=None):
        """All-gather of the two factors of a Linear weight gradient (dW = dy^T x): [B, In] and [B, Out] per rank become
        [world*B, ...].  For fc.1 that is 6.9 MB per rank on the wire instead of an 807 MB all-reduce, and the averaged
        gradient sum_r dy_r^T x_r / world is formed inside the fused wgrad + SGD kernel, never in memory.  Like the
        buckets it never blocks inside backward: RCCL gathers asynchronously and ``reduce`` waits; a host-waited backend
        on CUDA tensors keeps the local factors and ``reduce`` gathers them and stores the result in
        ``owner._yfs_factors`` (the returned tensors are then the local factors)."""
        if not self.active:
            return x16, dy16
        if x16.is_cuda and self._host_waited() and owner is not None:
            self._gathers.append((owner, x16, dy16))
            return x16, dy16
        outs = []
        for t in (x16, dy16):
            t = t.contiguous()
            out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            if not t.is_cuda or self._host_waited():          # CPU / gloo (tests), or no owner to patch later
                dist.all_gather(list(out.chunk(self.world)), t, group=self.group)
            else:
                self._works.append((dist.all_gather(list(out.chunk(self.world)), t, group=self.group, async_op=True), None, None))
            outs.append(out)
        return tuple(outs)

    def reduce(self, flat_g):
        """Issue whatever has not been issued yet (parameters that never reported), wait for everything (the host for a
        host-waited backend, else the current stream) and turn the sums into means."""
        if self.active:
            for idx in range(len(self.buckets)):
                if not self._issued[idx]:
                    self._issue(idx)
            inv = 1.0 / self.world
            if self._deferred or self._gathers:               # gloo on CUDA tensors: backward is fully enqueued by now
                cur = torch.cuda.current_stream()
                self._order_after_producers(cur)
                cur.synchronize()                             # this rank's own gradients are complete
                for owner, x16, dy16 in self._gathers:        # the same order on every rank
                    outs = []
                    for t in (x16, dy16):
                        t = t.contiguous()
                        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                        dist.all_gather(list(out.chunk(self.world)), t, group=self.group)
                        outs.append(out)
                    owner._yfs_factors = (outs[0], outs[1], inv)
                for idx in sorted(self._deferred):
                    lo, hi = self.buckets[idx]
                    chunk = flat_g[lo:hi]
                    buf = None
                    if self.compress:
                        buf = self._c16[idx]
                        buf.copy_(chunk)
                    self._works.append((dist.all_reduce(buf if buf is not None else chunk, op=dist.ReduceOp.SUM,
                                                        group=self.group, async_op=True), idx, buf))
                self._deferred, self._gathers = [], []
            for work, idx, buf in self._works:
                # nccl: the current stream waits for the collective (no host block); gloo: the host waits, and the
                # backend orders its copy back to the device before the stream that is current here
                if work is not None:
                    work.wait()
                if buf is not None:
                    lo, hi = self.buckets[idx]
                    torch.mul(buf, inv, out=flat_g[lo:hi])
            self._works = []
            if not self.compress:
                for lo, hi in self._scale_ranges:
                    flat_g[lo:hi].mul_(inv)
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
