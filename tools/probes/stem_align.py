#!/usr/bin/env python3
"""Timing-only experiment: stem forward(+pool) and pooled wgrad with the x rows 128-byte aligned (pixel stride
64 elements instead of 16; YFS_STEM_PS_EXPERIMENT=64 in the environment; results are garbage, the work is the same)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import conv_ops as co
dev = "cuda:0"
ps = int(os.environ.get("YFS_STEM_PS_EXPERIMENT", "16"))
n, hw, cout = 64, 448, 192
oh, ow, hs, ws = co.stem_geometry(hw, hw)
xs = torch.rand(n, hs, ws, ps, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)     # NHWC storage, logical NCHW
w2 = (torch.randn(cout, 256, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
yp, idx = co.stem_fwd_pool(xs, w2, b, oh, ow)
gp = torch.randn_like(yp)
def timed(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print(f"PS={ps}: fwd+pool {timed(lambda: co.stem_fwd_pool(xs, w2, b, oh, ow)):.1f} us, pooled wgrad {timed(lambda: co.stem_wgrad_pool(xs, gp, idx, None)):.1f} us")
