#!/usr/bin/env python3
"""Throughput of the mask -> boxes path (csrc/mask_kernels.hip) on VOC-sized masks (375 x 500, a few filled shapes + speckle),
with scipy.ndimage.label + find_objects on the host cores beside it (what the host version of this step costs).
    python tools/probes/mask_boxes_bench.py [batch] [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import data as yd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = 375, 500
rng = np.random.default_rng(0)
yy, xx = np.mgrid[:H, :W]
masks = np.zeros((N, H, W), np.uint8)
for i in range(N):
    m = (rng.uniform(size=(H, W)) < 0.002)
    for _ in range(6):
        cy, cx, ry, rx = rng.integers(0, H), rng.integers(0, W), rng.integers(3, H // 4), rng.integers(3, W // 4)
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    masks[i] = m
dev = torch.from_numpy(masks).cuda()
for _ in range(3):
    out = yd.masks_to_boxes(dev, max_boxes=1024)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = yd.masks_to_boxes(dev, max_boxes=1024)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
px = N * H * W
print(f"GPU: {N} masks of {H} x {W}: {ms * 1e3:.1f} us per batch = {N / ms * 1e3:.0f} masks/s, {px / ms / 1e6:.2f} Gpixel/s "
      f"({px / ms / 1e6:.2f} GB/s of mask bytes; components per mask: {out[1].float().mean().item():.0f})", flush=True)
from scipy import ndimage
k = min(N, 16)
t0 = time.perf_counter()
for i in range(k):
    lab, n = ndimage.label(masks[i], structure=np.ones((3, 3), np.int32))
    ndimage.find_objects(lab)
dt = (time.perf_counter() - t0) / k
print(f"host (scipy.ndimage.label + find_objects, 1 core): {dt * 1e3:.2f} ms per mask = {1 / dt:.0f} masks/s")
