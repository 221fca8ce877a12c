#!/usr/bin/env python3
"""Fused decode+NMS (one launch, boxes never leave LDS) against decode_grid followed by nms (two launches, the decoded
boxes/scores make a round trip through HBM) on BASELINE config 4, and at a larger batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth
import bench
dev = torch.device("cuda:0")
for N in (1024, 4096):
    raw = torch.from_numpy(synth.raw_head_v2(N, 13, 5, 20, seed=1234)).to(dev)
    anchors = torch.from_numpy(synth.VOC_ANCHORS).to(dev)
    op = y.DecodeNms(N, 13, 5, 20, dev)
    t_f = bench._time_ms(lambda: op(raw, anchors), 20)
    def unfused():
        b, s, c = y.decode_grid(raw, anchors)
        return y.nms(b, s, c, score_thr=0.05, iou_thr=0.5)
    t_u = bench._time_ms(unfused, 20)
    def dec():
        return y.decode_grid(raw, anchors)
    t_d = bench._time_ms(dec, 20)
    print(f"N={N}: fused {t_f:.4f} ms ({N / t_f / 1e3:.2f} M img/s) | decode_grid + nms {t_u:.4f} ms | decode_grid alone {t_d:.4f} ms", flush=True)
