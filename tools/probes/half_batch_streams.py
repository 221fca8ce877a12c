#!/usr/bin/env python3
"""Does running the backbone on batch slices in concurrent streams fill the partly empty last rounds of the 8-phase
kernels (196/392/784-tile grids on 256 CUs)?  Times model.feat forward (no_grad) and forward + backward for the full batch
on one stream against the same batch cut into 2 (or more) slices, one stream per slice.
    python tools/probes/half_batch_streams.py [batch] [slices ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd.model import YoloRegressorModel

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cuts = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
dev = torch.device("cuda")
torch.manual_seed(0)
model = YoloRegressorModel(head_out=1470).to(dev)
feat = model.feat
x = torch.rand(batch, 3, 448, 448, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
streams = [torch.cuda.Stream() for _ in range(max(cuts))]


def run(parts, backward):
    main = torch.cuda.current_stream()
    n = batch // parts
    outs = []
    ev0 = torch.cuda.Event()
    ev0.record(main)
    for p in range(parts):
        st = streams[p] if parts > 1 else main
        st.wait_event(ev0)
        with torch.cuda.stream(st):
            xs = x[p * n:(p + 1) * n]
            if backward:
                y = feat(xs)
                y.backward(torch.ones_like(y))
            else:
                with torch.no_grad():
                    y = feat(xs)
            outs.append(y)
    for p in range(parts):
        if parts > 1:
            main.wait_stream(streams[p])
    return outs


for backward in (False, True):
    for parts in cuts:
        for _ in range(3):
            run(parts, backward)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            run(parts, backward)
        e1.record()
        torch.cuda.synchronize()
        print(f"{'fwd+bwd' if backward else 'fwd    '}  batch {batch} in {parts} slice(s): {e0.elapsed_time(e1) / reps:8.3f} ms", flush=True)
