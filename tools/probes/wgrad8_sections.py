#!/usr/bin/env python3
"""Section clocks of every workgroup of wgrad8_kernel (-DYFS_IG8_TIMING build: make -C yolo_from_scratch_amd/csrc ig8-timing;
YFS_LIB_PATH=.../libyfs_ig8dbg.so): entry -> first MFMA, K-loop, epilogue, store drain, for the weight gradients of feat.10
(56 K-steps per workgroup), feat.18 (112), feat.24 (66) and feat.26 (49), batch 64."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


for name, cin, cout, hw in (("feat.10", 256, 512, 28), ("feat.18", 512, 1024, 28), ("feat.24", 1024, 1024, 14), ("feat.26", 1024, 1024, 7)):
    x = co.nhwc(torch.randn(64, cin, hw, hw, device="cuda"), torch.bfloat16)
    dz = co.nhwc(torch.randn(64, cout, hw, hw, device="cuda"), torch.bfloat16)
    for _ in range(3):
        co.conv_wgrad(x, dz, 3, 1)
    torch.cuda.synchronize()
    co.conv_wgrad(x, dz, 3, 1)
    torch.cuda.synchronize()
    kern = _lib.last_kernel()
    splits = int(kern.split("splits")[1].rstrip(">").split(">")[0]) if "splits" in kern else 1
    nblk = min(8192, (cout // 256) * (cin // 256) * 9 * splits)
    buf = (ctypes.c_longlong * (nblk * 6))()
    assert lib.yfs_debug_ig8_timeline(buf, nblk) == 0
    rows = [[buf[i * 6 + k] for k in range(5)] for i in range(nblk)]
    print(f"{name} weight gradient ({kern}): {nblk} workgroups")
    for lab, f in (("entry -> first MFMA", lambda t: t[1] - t[0]), ("K-loop", lambda t: t[2] - t[1]), ("epilogue", lambda t: t[3] - t[2]),
                   ("store drain", lambda t: t[4] - t[3])):
        v = [f(t) for t in rows]
        print(f"    {lab:20s} median {pct(v, 0.5):7d}  p10 {pct(v, 0.1):7d}  p90 {pct(v, 0.9):7d} cycles", flush=True)
