"""N>1 data-parallel gradient exchange (dp.GradComm) rehearsed on CPU: world_size 2, gloo, 127.0.0.1.
Checks: buckets tile the arena in reverse parameter order; bucket all-reduces fire when their last gradient
reports; the reduced arena equals the mean of the ranks' arenas == the single-rank gradient of the concatenated
batch; parameters broadcast from rank 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_from_scratch_amd import dp

SHAPES = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (128,), (1000, 300), (1000,), (10, 1000), (10,)]


def _layout():
    sizes = [(int(np.prod(s)) + 63) // 64 * 64 for s in SHAPES]
    offs, off = [], 0
    for s, sz in zip(SHAPES, sizes):
        offs.append((off, int(np.prod(s))))
        off += sz
    return offs, off


def test_bucket_layout_reverse_order_and_coverage():
    offs, total = _layout()
    params = [torch.nn.Parameter(torch.zeros(s)) for s in SHAPES]
    comm = dp.GradComm(world_size=2, bucket_bytes=256 * 1024)
    comm.attach(params, offs, torch.zeros(total))
    # first bucket holds the LAST parameters (fc first: they are produced first in backward)
    assert comm.param_bucket[id(params[-1])] == 0
    assert comm.param_bucket[id(params[0])] == len(comm.buckets) - 1
    covered = np.zeros(total, bool)
    for lo, hi in comm.buckets:
        assert not covered[lo:hi].any()
        covered[lo:hi] = True
    for off, n in offs:
        assert covered[off:off + n].all()
    assert len(comm.buckets) >= 2


def test_bucket_layout_skips_deferred_parameters():
    """A parameter whose gradient is never materialised (FlatSGD's deferred Linear weight) belongs to no bucket and
    no bucket spans its slice of the arena."""
    offs, total = _layout()
    params = [torch.nn.Parameter(torch.zeros(s)) for s in SHAPES]
    comm = dp.GradComm(world_size=2, bucket_bytes=1 << 30)            # one bucket if nothing were skipped
    skipped = params[4]                                                # the (1000, 300) "fc" weight
    comm.attach(params, offs, torch.zeros(total), skip_ids={id(skipped)})
    assert id(skipped) not in comm.param_bucket
    lo_s, n_s = offs[4]
    for lo, hi in comm.buckets:
        assert hi <= lo_s or lo >= lo_s + n_s
    assert len(comm.buckets) == 2
    for p, (off, n) in zip(params, offs):
        if p is not skipped:
            lo, hi = comm.buckets[comm.param_bucket[id(p)]]
            assert lo <= off and off + n <= hi


def test_scale_ranges_cover_the_buckets_and_nothing_of_a_skipped_parameter():
    """The 1/world factor is applied once per contiguous run of buckets (round 3: one launch per run instead of one per
    bucket).  The runs must cover every bucket element exactly once, may take in the < 64 zero elements of alignment
    padding between neighbours, and must not touch the slice of a parameter that has no bucket."""
    offs, total = _layout()
    params = [torch.nn.Parameter(torch.zeros(s)) for s in SHAPES]
    comm = dp.GradComm(world_size=2, bucket_bytes=64 * 1024)
    skipped = params[4]
    comm.attach(params, offs, torch.zeros(total), skip_ids={id(skipped)})
    assert len(comm.buckets) >= 3 and len(comm._scale_ranges) == 2          # before and after the skipped weight
    covered = np.zeros(total, int)
    for lo, hi in comm._scale_ranges:
        covered[lo:hi] += 1
    assert covered.max() == 1
    for lo, hi in comm.buckets:
        assert (covered[lo:hi] == 1).all()
    lo_s, n_s = offs[4]
    assert (covered[lo_s:lo_s + n_s] == 0).all()
    assert comm.exchange_streams == 0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs, total = _layout()
        params = [torch.nn.Parameter(torch.zeros(s)) for s in SHAPES]
        flat_p = torch.full((total,), float(rank + 1))
        dp.broadcast_parameters(flat_p)
        assert float(flat_p[0]) == 1.0 and float(flat_p[-1]) == 1.0
        flat_g = torch.zeros(total)
        comm = dp.GradComm(world_size=world, bucket_bytes=256 * 1024)
        comm.attach(params, offs, flat_g)
        for step in range(2):                       # two steps: per-step state must reset
            g = torch.Generator().manual_seed(100 * step + rank)
            local = torch.randn(total, generator=g)
            flat_g.copy_(local)
            issued_before = list(comm._issued)
            assert not any(issued_before)
            for p in reversed(params[2:]):          # backward order; the first two never report (unused params)
                comm.notify(p)
            assert comm._issued[0]                  # the fc bucket went out before the end of "backward"
            comm.reduce(flat_g)
            others = [torch.randn(total, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
            expect = sum(others) / world
            for off, n in offs:
                np.testing.assert_allclose(flat_g[off:off + n].numpy(), expect[off:off + n].numpy(), rtol=1e-6, atol=1e-7)
        # factors of a deferred Linear gradient: gathered in rank order, so sum_r dy_r^T x_r = DY^T X
        x = torch.full((3, 8), float(rank + 1)).bfloat16()
        dy = torch.full((3, 4), float(10 * (rank + 1))).bfloat16()
        xa, da = comm.gather_factors(x, dy)
        assert xa.shape == (3 * world, 8) and da.shape == (3 * world, 4)
        for r in range(world):
            assert float(xa[3 * r, 0]) == r + 1 and float(da[3 * r + 2, 3]) == 10 * (r + 1)
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_grad_exchange_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
