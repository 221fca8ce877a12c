#!/usr/bin/env python3
"""The dgrad of the block behind a pool, carried through the pool in its epilogue (yfs_conv2d_dgrad_unpool_bf16), against the
two launches it replaces, at the three places the backbone has it (feat.4, feat.9, feat.20; batch 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import conv_ops as co, model as ym, _lib


def time_us(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, cin, cout, hw in (("feat.4", 256, 128, 56), ("feat.9", 512, 256, 28), ("feat.20", 1024, 512, 14)):
    n = 64
    x_full = co.nhwc(torch.randn(n, cin, 2 * hw, 2 * hw, device="cuda"), torch.bfloat16)
    y, idx = ym._pool_fwd_idx(x_full)
    del x_full
    dz = co.nhwc(torch.randn(n, cout, hw, hw, device="cuda"), torch.bfloat16)
    w = co.nhwc((torch.rand(cout, cin, 1, 1, device="cuda") * 2 - 1) / cin ** 0.5)
    wt16 = co.weight_prep(w)[1]
    t_d = time_us(lambda: co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=y))
    kd = _lib.last_kernel()
    g = co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=y)
    t_p = time_us(lambda: ym._pool_bwd_idx(y, idx, g, fuse_leaky=False))
    t_f = time_us(lambda: co.conv_dgrad_unpool(dz, wt16, y, idx))
    kf = _lib.last_kernel()
    print(f"{name}: dgrad {t_d:6.1f} us ({kd}) + pool backward {t_p:6.1f} us = {t_d + t_p:6.1f} us; one launch {t_f:6.1f} us ({kf})")
