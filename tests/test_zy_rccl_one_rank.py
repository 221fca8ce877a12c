"""The data-parallel exchange (dp.GradComm) on REAL RCCL, in a process group of one rank.

A one-GPU box cannot hold two RCCL ranks, so the N > 1 path is otherwise covered by gloo (CPU tensors in
tests/test_dp_gloo.py, CUDA tensors in tools/dp_gpu_check.py) and by faked collectives (tests/test_gpu_step.py).  What
none of those execute is the RCCL branch of ``GradComm._issue`` / ``gather_factors`` / ``broadcast_parameters``: the
asynchronous ``all_reduce`` enqueued behind the exchange stream, ``work.wait()`` as a stream dependency, ``all_gather``
into chunk views of one output tensor on a side stream, the bf16 staging buffer, ``init_process_group("nccl", device_id=...)``
as bench.py calls it.  With ``force=True`` the exchange runs in full in a group of one rank, where it must change nothing:
the trajectory has to equal the run without exchange bit for bit (fp32 buckets; the mean over one rank is the value itself).
Collected after every other test file except the no-allocation one, and skipped (not failed) when the box cannot create
the group at all."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import dp, synth
from yolo_from_scratch_amd import model as ym
from yolo_from_scratch_amd import train as yt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rccl_group():
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        probe = torch.ones(4, device=DEV)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001  (an environment that cannot start RCCL is not a defect of this package)
        if dist.is_initialized():
            dist.destroy_process_group()
        pytest.skip(f"cannot create a one-rank RCCL group on this box: {e!r}")
    assert float(probe.sum()) == 4.0
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _run(comm, steps=3, size=64, batch=4, lr=1e-2):
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(7)).to(DEV)
    tgt = torch.from_numpy(synth.v1_targets(batch, seed=3)).to(DEV)
    crit = y.YoloV1Loss(7, 2, 20)
    torch.manual_seed(0)
    net = ym.YoloRegressorModel(head_out=1470, image_size=size, p_drop=0.0).to(DEV)
    net.fc[1].weight._yfs_defer_capable = True          # the deferred (factor-gathering) fc.1 path at this small size too
    opt = yt.FlatSGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4, max_norm=1.0, comm=comm)
    dp.broadcast_parameters(opt)
    for _ in range(steps):
        opt.zero_grad()
        crit(net(x), tgt).backward()
        opt.step()
    torch.cuda.synchronize()
    return opt


def test_exchange_on_rccl_in_a_one_rank_group_changes_nothing(rccl_group):
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    base = _run(None)
    comm = dp.GradComm(bucket_bytes=36 << 20, tail_bytes=16 << 20, tail_bucket_bytes=4 << 20, force=True)
    assert comm.world == 1 and comm.active
    got = _run(comm)
    assert len(comm.buckets) >= 8 and any(e - s < (2 << 20) for s, e in comm.buckets)
    assert comm.exchange_streams == 0                    # the collectives ride on RCCL's own stream: none added here
    assert torch.equal(got.flat_p, base.flat_p) and torch.equal(got.flat_m, base.flat_m)
    assert torch.equal(got.flat_p16, base.flat_p16)
    # bf16 buckets: the gradients are rounded to bf16 on the wire, nothing else may differ
    got16 = _run(dp.GradComm(bucket_bytes=8 << 20, compress_bf16=True, force=True))
    assert torch.isfinite(got16.flat_p).all() and _rel(got16.flat_p, base.flat_p) < 1e-4


def test_full_size_step_with_rccl_exchange(rccl_group):
    """BASELINE configs[1] shapes (448 x 448, 1470 outputs, fc.1 = 50176 -> 4096 with its 6.9 MB of gathered factors and the
    ~36 MB conv buckets), batch 8: the step with the forced exchange equals the step without it."""
    base = _run(None, steps=2, size=448, batch=8, lr=1e-3)
    got = _run(dp.GradComm(force=True), steps=2, size=448, batch=8, lr=1e-3)
    assert torch.equal(got.flat_p, base.flat_p) and torch.equal(got.flat_m, base.flat_m)
