"""Weight-gradient split-K factor: the cost model's choice against its neighbours, per layer (same process, HIP events).
    python tools/probes/wgrad_split_sweep.py [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_from_scratch_amd import _lib as L, conv_ops as co
from tools.conv_bench import LAYERS

dev = "cuda:0"
layers = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,7,10,18,21,24,25,26").split(",")]


def timed(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


for li in layers:
    cin, cout, k, s, hw = LAYERS[li]
    oh, ow = co.out_hw(hw, hw, k, s)
    x = torch.randn(64, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
    dz = torch.randn(64, cout, oh, ow, device=dev).to(torch.bfloat16).contiguous(memory_format=co.CL)
    fn = lambda: co.conv_wgrad(x, dz, k, s)
    L.set_knob("wgrad_splits", 0)
    fn()
    name = L.last_kernel()
    chosen = int(name.split("splits")[1].rstrip(">").split(">")[0].split(",")[0])
    t0 = timed(fn)
    line = f"L{li:<2d} {name:40s} {t0:7.1f} us |"
    cands = sorted({max(1, int(round(chosen * f))) for f in (0.5, 0.67, 0.8, 0.9, 1.1, 1.25, 1.5, 2.0, 3.0, 4.0, 6.0)} - {chosen})
    for sp in cands:
        L.set_knob("wgrad_splits", sp)
        try:
            fn()
            real = L.last_kernel().split("splits")[1].rstrip(">")
            line += f" {real}:{timed(fn):6.1f}"
        finally:
            L.set_knob("wgrad_splits", 0)
    print(line, flush=True)
