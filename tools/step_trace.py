#!/usr/bin/env python3
"""Per-step wall time of the bench train step (debug aid): python tools/step_trace.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth, train as yt, conv_ops as co
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
from yolo_from_scratch_amd.model import YoloRegressorModel
model = YoloRegressorModel(head_out=1470).to(dev)
opt = yt.FlatSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.0)
crit = y.YoloV1Loss(7, 2, 20)
x = torch.rand(64, 3, 448, 448).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
target = torch.from_numpy(synth.v1_targets(64, seed=1)).to(dev)
wg = []
orig = co.conv_wgrad
def wrapped(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(*a, **k); e1.record(); wg.append((e0, e1, a[0].shape, a[1].shape)); return out
co.conv_wgrad = wrapped
ev = []
import gc
if os.environ.get('NOGC'):
    gc.collect(); gc.disable()
for i in range(steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    opt.zero_grad(); out = model(x); loss = crit(out, target); loss.backward(); opt.step()
    e1.record(); ev.append((e0, e1, loss))
torch.cuda.synchronize()
per = len(wg) // steps
for i, (a, b, l) in enumerate(ev):
    w = wg[i * per:(i + 1) * per]
    ws = sorted(((p.elapsed_time(q), tuple(s1), tuple(s2)) for p, q, s1, s2 in w), reverse=True)[:2]
    print(i, f"{a.elapsed_time(b):7.2f} ms  loss {float(l):8.4f}  wgrad {sum(p.elapsed_time(q) for p, q, _, _ in w):6.2f}", [(round(t, 2), s1[1], s2[1], s1[2]) for t, s1, s2 in ws])
