#!/usr/bin/env python3
"""A launch of igemm8_kernel that is nearly all epilogue (1x1 convolution over 64 input channels: ONE K-step per tile), for
rocprofv3 --pmc runs: LDS bank conflicts, LDS / VALU instruction counts and busy cycles of the epilogue by itself."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import _lib, conv_ops as co
_lib.set_knob("igemm_tile", 87)
cin, cout, hw, batch = 64, 256, 56, 64
x = co.nhwc((torch.rand(batch, cin, hw, hw, device="cuda") - 0.3), torch.bfloat16)
w = co.nhwc((torch.rand(cout, cin, 1, 1, device="cuda") * 2 - 1) / cin ** 0.5)
b = torch.zeros(cout, device="cuda")
w16, _ = co.weight_prep(w, want_wt16=False)
for _ in range(5):
    y = co.conv_fwd(x, w16, b, 1)
torch.cuda.synchronize()
print(_lib.last_kernel())
