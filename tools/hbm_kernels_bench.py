#!/usr/bin/env python3
"""The three HBM-bound sub-benchmarks of bench.py on their own (for rocprofv3 --pmc runs):
    python tools/hbm_kernels_bench.py [nms] [loss] [assign]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

d = torch.device("cuda:0")
which = sys.argv[1:] or ["nms", "loss", "assign"]
for name in which:
    r = getattr(bench, name + "_bench")(d)
    print(name, r["ms_per_batch"], "ms", r["roofline"]["achieved"], "GB/s", r["roofline"]["frac"], flush=True)
