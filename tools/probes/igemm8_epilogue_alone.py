#!/usr/bin/env python3
"""Is the epilogue of igemm8_kernel bound by all CUs writing at once, or by its own LDS round trip?  Section clocks of
workgroup 17 (libyfs_ig8dbg.so: make -C yolo_from_scratch_amd/csrc ig8-timing; run with YFS_LIB_PATH=.../libyfs_ig8dbg.so)
for feat.18's forward and masked dgrad with 32 workgroups on the chip (batch 2), one full round (batch 16) and batch 64."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()
co.SPLIT_K = False
_lib.set_knob("igemm_tile", 87)
cin, cout, hw = 512, 1024, 28
w = co.nhwc((torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) / (cin * 9) ** 0.5)
b = torch.zeros(cout, device="cuda")
w16, wt16 = co.weight_prep(w)
for batch in (2, 16, 64):
    x = co.nhwc((torch.rand(batch, cin, hw, hw, device="cuda") - 0.3), torch.bfloat16)
    dz = co.nhwc(torch.randn(batch, cout, hw, hw, device="cuda"), torch.bfloat16)
    for op, fn in (("forward", lambda: co.conv_fwd(x, w16, b, 1)), ("masked dgrad", lambda: co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=x))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 8)()
        assert lib.yfs_debug_ig8_times(buf) == 0
        t = [buf[i] for i in range(5)]
        print(f"batch {batch:2d} {op:13s} ({_lib.last_kernel()}): fill {t[1] - t[0]} cyc, loop {t[2] - t[1]}, epilogue {t[3] - t[2]}, "
              f"store drain {t[4] - t[3]}", flush=True)
