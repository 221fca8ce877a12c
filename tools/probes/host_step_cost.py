#!/usr/bin/env python3
"""Host time of one train step: the same launch sequence as the bench step, on an input so small (batch 1, 64 x 64) that
the GPU work is negligible, so the wall time per step is what the host needs to enqueue it.  With YFS_FORCE_DP=1 the
gradient exchange runs in a one-rank RCCL group (dp.GradComm(force=True)): what does issuing the buckets cost the host?
    python tools/probes/host_step_cost.py [steps]"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import dp, synth, model as ym, train as yt

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
force = os.environ.get("YFS_FORCE_DP") == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
# the real model's layer list and parameter sizes (so buckets are the real ones), fc.1 sized for 64 x 64
net = ym.YoloRegressorModel(head_out=1470, image_size=size).to(dev)
net.fc[1].weight._yfs_defer_capable = True
comm = dp.GradComm(force=True) if force else None
opt = yt.FlatSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.0, comm=comm)
x = torch.rand(batch, 3, size, size, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
tgt = torch.from_numpy(synth.v1_targets(batch, seed=1)).to(dev)
crit = y.YoloV1Loss(7, 2, 20)


def step():
    opt.zero_grad(set_to_none=True)
    crit(net(x), tgt).backward()
    opt.step()


import gc
for _ in range(10):
    step()
torch.cuda.synchronize(); gc.collect(); gc.disable()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
parts = [0.0, 0.0, 0.0]
for _ in range(steps):
    a = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x), tgt)
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step()
    d = time.perf_counter()
    parts[0] += b - a; parts[1] += c - b; parts[2] += d - c
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{'forced one-rank RCCL exchange' if force else 'no exchange':32s} host {1e3 * (t1 - t0) / steps:6.3f} ms/step to enqueue, "
      f"(forward {1e3 * parts[0] / steps:.3f}, backward {1e3 * parts[1] / steps:.3f}, optimiser {1e3 * parts[2] / steps:.3f}), "
      f"{1e3 * (t2 - t0) / steps:6.3f} ms/step until the GPU is done, GPU span {e0.elapsed_time(e1) / steps:.3f} ms/step" + (f"  ({len(comm.buckets)} buckets)" if comm else ""), flush=True)
if force:
    dist.destroy_process_group()
