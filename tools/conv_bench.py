#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernel set at BASELINE config 2 shapes (batch 64, 448x448).
    python tools/conv_bench.py [--ops fwd,dgrad,wgrad | step] [--layers 2,7,18] [--iters 10] [--batch 64]
Prints TFLOP/s per layer and op (HIP-event timing, random bf16 data).  --ops step: per layer exactly the launches the
training step makes (pool-fused forward for feat.0/2/7/18, the stem's weight gradient from the pooled side, no dgrad
for the stem), so a profile of this run shows the step's kernels and nothing else."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_from_scratch_amd import conv_ops as co   # noqa: E402

# feat index -> (cin, cout, k, stride, in_hw)
LAYERS = {0: (3, 192, 7, 2, 448),      # stem: space-to-depth + 4x4 implicit GEMM (useful FLOPs quoted)
          2: (192, 256, 3, 1, 112), 4: (256, 128, 1, 1, 56), 5: (128, 256, 3, 1, 56), 6: (256, 256, 1, 1, 56),
          7: (256, 512, 3, 1, 56), 9: (512, 256, 1, 1, 28), 10: (256, 512, 3, 1, 28), 17: (512, 512, 1, 1, 28),
          18: (512, 1024, 3, 1, 28), 20: (1024, 512, 1, 1, 14), 21: (512, 1024, 3, 1, 14), 24: (1024, 1024, 3, 1, 14),
          25: (1024, 1024, 3, 2, 14), 26: (1024, 1024, 3, 1, 7)}
COUNT = {0: 1, 2: 1, 4: 1, 5: 1, 6: 1, 7: 1, 9: 4, 10: 4, 17: 1, 18: 1, 20: 2, 21: 2, 24: 1, 25: 1, 26: 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default=",".join(str(k) for k in LAYERS))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--igemm-tile", type=int, default=0, help="YFS_KNOB_IGEMM_TILE (0 = library's choice)")
    ap.add_argument("--wgrad-tile", type=int, default=0, help="YFS_KNOB_WGRAD_TILE")
    ap.add_argument("--wgrad-splits", type=int, default=0, help="YFS_KNOB_WGRAD_SPLITS")
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    L.set_knob("igemm_tile", a.igemm_tile)
    L.set_knob("wgrad_tile", a.wgrad_tile)
    L.set_knob("wgrad_splits", a.wgrad_splits)
    dev = "cuda:0"
    step_ops = a.ops == "step"
    if step_ops:
        a.ops = "fwd,fwdpool,dgrad,wgrad,wgradpool"
    POOLED = (0, 2, 7, 18)
    tot = {op: 0.0 for op in a.ops.split(",")}
    for li in [int(v) for v in a.layers.split(",")]:
        cin, cout, k, s, hw = LAYERS[li]
        oh, ow = co.out_hw(hw, hw, k, s)
        if li == 0:
            x = torch.rand(a.batch, 3, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
            w = (torch.randn(cout, 3, 7, 7, device=dev) * 0.05).contiguous(memory_format=co.CL)
            b = torch.randn(cout, device=dev)
            dz = torch.randn(a.batch, cout, oh, ow, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
            xs = co.stem_space_to_depth(x)
            w2 = co.stem_weight(w)
            gp = torch.randn(a.batch, cout, oh // 2, ow // 2, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
            yp, pidx = co.stem_fwd_pool(xs, w2, b, oh, ow)
            flops = 2.0 * a.batch * oh * ow * cout * 147
            line = f"L0     3-> 192 k7 s2 448^2  GF={flops / 1e9:7.1f}"
            for op, fn in (("s2d", lambda: co.stem_space_to_depth(x)), ("fwd", lambda: co.stem_fwd(xs, w2, b, oh, ow)),
                           ("fwdpool", lambda: co.stem_fwd_pool(xs, w2, b, oh, ow)), ("wgrad", lambda: co.stem_wgrad(xs, dz)),
                           ("wgradpool", lambda: co.stem_wgrad_pool(xs, gp, pidx, yp))):
                if op != "s2d" and op not in tot:
                    continue
                if step_ops and op in ("fwd", "wgrad"):
                    continue
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / a.iters * 1e3
                if op in tot:
                    tot[op] += us
                line += f" | {op} {us:8.1f} us {flops / us / 1e6:7.1f} TF"
            print(line, flush=True)
            continue
        x = torch.randn(a.batch, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=co.CL)
        b = torch.randn(cout, device=dev)
        w16, wt16 = co.weight_prep(w)
        dz = torch.randn(a.batch, cout, oh, ow, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
        flops = 2.0 * a.batch * oh * ow * cout * cin * k * k
        line = f"L{li:<2d} {cin:4d}->{cout:4d} k{k} s{s} {hw:3d}^2  GF={flops / 1e9:7.1f}"
        for op in a.ops.split(","):
            if op == "wgradpool" or (op == "fwdpool" and oh % 2):
                continue          # stem only / odd output: no fused pool
            if step_ops and op == ("fwd" if li in POOLED else "fwdpool"):
                continue
            fn = {"fwd": lambda: co.conv_fwd(x, w16, b, s), "fwdpool": lambda: co.conv_fwd_pool(x, w16, b, s),
                  "dgrad": lambda: co.conv_dgrad(dz, wt16, (hw, hw), s, x_act=x),
                  "wgrad": lambda: co.conv_wgrad(x, dz, k, s)}[op]
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            tot[op] += us * COUNT[li]
            line += f" | {op} {us:8.1f} us {flops / us / 1e6:7.1f} TF"
        print(line, flush=True)
    print("per-step totals (us):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))


if __name__ == "__main__":
    main()
