#!/usr/bin/env python3
"""Fixed and per-K-step cost of the 8-phase forward kernel: 3x3 convolutions at 28^2 (batch 64, Cout = 1024: 224 x 4 = 896
tiles of 224 rows = 3.5 rounds) and at 14^2 (56 x 4 = 224 tiles = one round at 87.5 %) with Cin = 256 ... 2048, i.e. 36 ... 288
K-steps per tile.  time = rounds * (a + b * ksteps): prints the least-squares a (us per round) and b (us per K-step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import conv_ops as co, _lib as L
import numpy as np
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.set_knob("igemm_tile", tile)
for hw, rounds in ((28, 4), (14, 1)):
    pts = []
    for cin in (256, 512, 1024, 2048):
        x = co.nhwc((torch.rand(64, cin, hw, hw, device="cuda") - 0.3), torch.bfloat16)
        w = co.nhwc((torch.rand(1024, cin, 3, 3, device="cuda") * 2 - 1) / (cin * 9) ** 0.5)
        b = torch.zeros(1024, device="cuda")
        w16, _ = co.weight_prep(w, want_wt16=False)
        for _ in range(3):
            co.conv_fwd(x, w16, b, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            co.conv_fwd(x, w16, b, 1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        ks = 9 * cin // 64
        pts.append((ks, us))
        print(f"hw {hw} cin {cin:5d}  ksteps {ks:4d}  {us:8.1f} us  {2 * 64 * hw * hw * 1024 * 9 * cin / us / 1e6:7.1f} TF")
    A = np.array([[rounds, rounds * k] for k, _ in pts], float)
    y = np.array([u for _, u in pts])
    (a, bb), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"  hw {hw}: a = {a:.1f} us per round, b = {bb:.3f} us per K-step (rounds = {rounds})")
