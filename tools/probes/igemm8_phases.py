#!/usr/bin/env python3
"""Section clocks of one igemm8_kernel workgroup (needs libyfs_ig8dbg.so: `make -C yolo_from_scratch_amd/csrc ig8-timing`).
Run as  YFS_LIB_PATH=yolo_from_scratch_amd/libyfs_ig8dbg.so python tools/probes/igemm8_phases.py
For a few 3x3 layers (batch 64): cycles of workgroup 17 from entry to the first barrier after the prologue (pipeline fill), in
the main loop, in the epilogue up to its last store issue, and until those stores have left (vmcnt(0))."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()
for name, cin, cout, hw in (("L7", 256, 512, 56), ("L10", 256, 512, 28), ("L18", 512, 1024, 28), ("L24", 1024, 1024, 14)):
    x = co.nhwc((torch.rand(64, cin, hw, hw, device="cuda") - 0.3), torch.bfloat16)
    w = co.nhwc((torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) / (cin * 9) ** 0.5)
    b = torch.zeros(cout, device="cuda")
    w16, _ = co.weight_prep(w, want_wt16=False)
    for _ in range(3):
        co.conv_fwd(x, w16, b, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); co.conv_fwd(x, w16, b, 1); e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 8)()
    assert lib.yfs_debug_ig8_times(buf) == 0
    t = [buf[i] for i in range(5)]
    ks = 9 * cin // 64
    tot = t[4] - t[0]
    print(f"{name}: launch {e0.elapsed_time(e1) * 1e3:7.1f} us, {ks} K-steps; workgroup 17: fill {t[1] - t[0]} cyc, loop {t[2] - t[1]} "
          f"({(t[2] - t[1]) / ks:.0f}/K-step), epilogue {t[3] - t[2]}, store drain {t[4] - t[3]}, total {tot}")
