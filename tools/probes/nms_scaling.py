#!/usr/bin/env python3
"""NMS kernel time against the number of images (config 4 data): tells a latency-bound workgroup (time flat up to one
resident round = 4 workgroups per CU = 1024 images) from an issue-bound one (time grows with the images per CU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth
S, A, C = 13, 5, 20
anc = torch.from_numpy(synth.VOC_ANCHORS[:A]).cuda()
for N in (64, 128, 256, 512, 1024, 2048, 4096):
    raw = torch.from_numpy(synth.raw_head_v2(N, S, A, C, seed=1234)).cuda()
    boxes, score, cls = y.decode_grid(raw, anc)
    nb = S * S * A
    boxes = boxes.reshape(N, nb, 4).contiguous(); score = score.reshape(N, nb).contiguous(); cls = cls.reshape(N, nb).contiguous()
    for _ in range(3):
        keep, count = y.nms(boxes, score, cls, 0.05, 0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.nms(boxes, score, cls, 0.05, 0.5)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    e0.record()
    for _ in range(20):
        y.decode_grid(raw, anc)
    e1.record(); torch.cuda.synchronize()
    md = e0.elapsed_time(e1) / 20
    print(f"N={N:5d}  nms {ms * 1e3:7.1f} us  ({ms * 1e3 / N * 1024:6.1f} us per 1024)   decode {md * 1e3:6.1f} us   kept/img {float(count.float().mean()):.1f}")
