#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one YOLOv1-448 train step (S=7, B=2, C=20, per-GPU batch 64, bf16 conv on MFMA): zero_grad,
forward (24 Conv2D blocks + 4 pools + fc head), fused composite loss, backward, RCCL gradient exchange
(N > 1), clip_grad_norm_(1.0) + SGD(momentum 0.9, wd 5e-4).  Inputs are resident in HBM before the timed
region.  `value` = global images/s.  The JSON line also carries:
  roofline      the conv kernel set (MFMA-bound), achieved = algorithmic FLOPs / HIP-event time of the conv launches
                inside the timed region; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference step, oracle/model_ref.py) timed on the
                host cores, rank 0, N = 1 only, bounded sample
  nms           BASELINE config 4 (decode + IoU + class-wise NMS, 1024 images x 845 boxes): images/s and HBM roofline
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same table
BATCH = 64
NMS_BYTES_PER_IMG = 108160    # raw head 84,500 + boxes/score/class 20,280 + kept indices 3,380


def conv_flops_table(model, batch, hw=448):
    """Algorithmic FLOPs per conv launch (2*MACs), keyed like the timers in conv_ops."""
    from yolo_from_scratch_amd import model as ym
    from yolo_from_scratch_amd import conv_ops as co
    fwd = dgrad = wgrad = 0.0
    h = hw
    first = True
    for m in model.feat:
        if isinstance(m, ym.HipMaxPool2d):
            h //= 2
            continue
        c = m[0]
        oh, _ = co.out_hw(h, h, c.kernel_size, c.stride)
        macs = batch * oh * oh * c.out_channels * c.in_channels * c.kernel_size ** 2
        fwd += 2 * macs
        wgrad += 2 * macs
        if not first:
            dgrad += 2 * macs
        first = False
        h = oh
    return {"conv_fwd": fwd, "conv_dgrad": dgrad, "conv_wgrad": wgrad}


class ConvTimers:
    """HIP-event brackets around every conv launch (they run on torch's current stream)."""

    def __init__(self):
        self.events = {"conv_fwd": [], "conv_dgrad": [], "conv_wgrad": []}

    def install(self):
        from yolo_from_scratch_amd import conv_ops as co
        self._orig = {}
        for key, name in (("conv_fwd", "conv_fwd"), ("conv_dgrad", "conv_dgrad"), ("conv_wgrad", "conv_wgrad"),
                          ("conv_fwd", "stem_fwd"), ("conv_wgrad", "stem_wgrad")):
            fn = getattr(co, name)
            self._orig[name] = fn

            def wrapped(*a, _fn=fn, _key=key, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _fn(*a, **kw)
                e1.record()
                self.events[_key].append((e0, e1))
                return out
            setattr(co, name, wrapped)

    def remove(self):
        from yolo_from_scratch_amd import conv_ops as co
        for name, fn in self._orig.items():
            setattr(co, name, fn)

    def totals_ms(self):
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.events.items()}


def cpu_baseline_train(sample_batch=4, steps=2):
    """The oracle's train step (torch-CPU fp32 restatement of the reference step + C composite loss) on the host."""
    import oracle
    from oracle import model_ref
    from yolo_from_scratch_amd import synth
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    torch.set_num_threads(cores)
    params = model_ref.init_params(seed=0, head_out=1470)
    bufs = {}
    x = torch.rand(sample_batch, 3, 448, 448, generator=torch.Generator().manual_seed(0))
    tgt = synth.v1_targets(sample_batch, seed=1)

    class _OracleLoss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, out):
            lo, g = oracle.yolov1_loss(out.detach().numpy().reshape(-1, 7, 7, 30), tgt)
            ctx.g = torch.from_numpy(g.reshape(out.shape))
            return torch.tensor(float(lo[0]))

        @staticmethod
        def backward(ctx, go):
            return ctx.g * go

    loss_fn = lambda out, _t: _OracleLoss.apply(out)   # noqa: E731
    model_ref.train_step(params, bufs, x, None, loss_fn)          # warm-up (allocations, oneDNN primitive cache)
    t0 = time.perf_counter()
    for _ in range(steps):
        model_ref.train_step(params, bufs, x, None, loss_fn)
    dt = time.perf_counter() - t0
    return {"value": round(sample_batch * steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps at batch {sample_batch}, 448x448, fp32 torch-CPU restatement "
                      f"(oracle/model_ref.py + oracle yolov1_loss), 1 warm-up step untimed"}


def nms_bench(device, iters=20):
    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import synth
    N = 1024
    raw = torch.from_numpy(synth.raw_head_v2(N, 13, 5, 20, seed=1234)).to(device)
    anchors = torch.from_numpy(synth.VOC_ANCHORS).to(device)
    op = y.DecodeNms(N, 13, 5, 20, device)
    for _ in range(3):
        op(raw, anchors)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op(raw, anchors)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gbs = N * NMS_BYTES_PER_IMG / (ms * 1e-3) / 1e9
    out = {"metric": "NMS images/sec (decode + IoU + class-wise NMS, 13x13x5 anchors, C=20, batch 1024)",
           "value": round(N / (ms * 1e-3), 1), "unit": "images/s", "ms_per_batch": round(ms, 4), "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 5), "traffic": None}}
    return out


def cpu_baseline_nms(sample=64):
    import oracle
    from yolo_from_scratch_amd import synth
    raw = synth.raw_head_v2(sample, 13, 5, 20, seed=1234)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        b, s, c = oracle.decode_grid(raw, synth.VOC_ANCHORS)
        oracle.nms(b, s, c, 0.05, 0.5)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample * reps / dt, 1), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes over {sample} images of config 4 (C oracle, scalar)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--compress-bf16", action="store_true", help="bf16 gradient buckets for the all-reduce")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import dp, synth
    from yolo_from_scratch_amd import model as ym
    from yolo_from_scratch_amd import train as yt

    torch.manual_seed(0)
    model = ym.YoloRegressorModel(head_out=7 * 7 * 30).to(device)
    model.train()                                    # Dropout(0.2) active, as in the reference's train loop
    comm = dp.GradComm(world, compress_bf16=args.compress_bf16) if world > 1 else None
    opt = yt.FlatSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.0, comm=comm)
    dp.broadcast_parameters(opt)
    crit = y.YoloV1Loss(7, 2, 20)
    g = torch.Generator().manual_seed(rank)
    x = torch.rand(args.batch, 3, 448, 448, generator=g).to(device).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    target = torch.from_numpy(synth.v1_targets(args.batch, seed=1 + rank)).to(device)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = crit(out, target)
        loss.backward()
        opt.step()
        return loss

    # A full (generation-2) Python garbage collection walks every object torch created at import: ~75 ms, which
    # the GPU spends idle.  It fired once every ~30 steps and cost 2-4 ms/step in short runs; the timed loop
    # allocates no reference cycles, so collect once and keep the collector off while timing.
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    timers = ConvTimers()
    timers.install()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    timers.remove()
    final_loss = float(loss)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    if rank == 0:
        flops = conv_flops_table(model, args.batch)
        tot = timers.totals_ms()
        conv_ms = sum(v[0] for v in tot.values()) / args.steps
        conv_flops = sum(flops.values())
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12
        kernels = {k: {"launches_per_step": tot[k][1] // args.steps, "ms_per_step": round(tot[k][0] / args.steps, 4),
                       "tflops": round(flops[k] / (tot[k][0] / args.steps * 1e-3) / 1e12, 1)} for k in tot}
        rec = {
            "metric": "train images/sec YOLOv1-448", "value": round(args.batch * world * args.steps / dt, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "YOLOv1 448x448 S=7 B=2 C=20 train step, per-GPU batch %d, bf16 conv MFMA + fused "
                                   "composite loss + clip + SGD (BASELINE configs[1]; configs[2] when n_gpus=8)" % args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "grad_allreduce": ("bf16" if args.compress_bf16 else "fp32") if world > 1 else None},
            "final_loss": round(final_loss, 5),
            "roofline": {"bound": "mfma", "kernel": "conv implicit-GEMM set (igemm_kernel fwd+dgrad, wgrad_kernel)",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                         "algorithmic_tflop_per_step": round(conv_flops / 1e12, 4), "conv_ms_per_step": round(conv_ms, 3),
                         "kernels": kernels},
        }
        if world == 1 and not args.no_nms:
            rec["nms"] = nms_bench(device)
            if not args.no_cpu_baseline:
                rec["nms"]["cpu_baseline"] = cpu_baseline_nms()
        if world == 1 and not args.no_cpu_baseline:
            del model, opt
            torch.cuda.empty_cache()
            rec["cpu_baseline"] = cpu_baseline_train()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
