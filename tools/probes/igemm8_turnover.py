#!/usr/bin/env python3
"""What lies between two workgroups of igemm8_kernel on one CU?  The -DYFS_IG8_TIMING build (make -C yolo_from_scratch_amd/csrc
ig8-timing; YFS_LIB_PATH=.../libyfs_ig8dbg.so) records the section clocks and the hardware id of every workgroup; this lines the
workgroups of each CU up in time and prints the distribution of (next start - previous end), of each section, and the share
of a CU's busy span that is the K-loop.  feat.7's masked dgrad and forward (1792 tiles = 7 rounds), batch 64."""
import ctypes, os, sys, collections, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()
co.SPLIT_K = False


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


for name, cin, cout, hw in (("feat.7", 256, 512, 56), ("feat.18", 512, 1024, 28)):
    batch = 64
    w = co.nhwc((torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) / (cin * 9) ** 0.5)
    b = torch.zeros(cout, device="cuda")
    w16, wt16 = co.weight_prep(w)
    x = co.nhwc((torch.rand(batch, cin, hw, hw, device="cuda") - 0.3), torch.bfloat16)
    dz = co.nhwc(torch.randn(batch, cout, hw, hw, device="cuda"), torch.bfloat16)
    for op, fn in (("forward", lambda: co.conv_fwd(x, w16, b, 1)), ("masked dgrad", lambda: co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=x))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        kern = _lib.last_kernel()
        rows = 224 if "224" in kern else 256
        nout = cout if op == "forward" else cin
        nblk = -(-batch * hw * hw // rows) * (nout // 256)
        buf = (ctypes.c_longlong * (nblk * 6))()
        assert lib.yfs_debug_ig8_timeline(buf, nblk) == 0
        per_cu = collections.defaultdict(list)
        for i in range(nblk):
            t = [buf[i * 6 + k] for k in range(6)]
            hw_id, xcc = t[5] & 0xFFFFFFFF, (t[5] >> 32) & 0xF
            cu = (xcc, (hw_id >> 13) & 7, (hw_id >> 12) & 1, (hw_id >> 8) & 15)   # xcc, se, sh, cu
            per_cu[cu].append(t)
        gaps, fills, loops, epis, drains, share = [], [], [], [], [], []
        for cu, wl in per_cu.items():
            wl.sort(key=lambda t: t[0])
            for a, bb in zip(wl, wl[1:]):
                gaps.append(bb[0] - a[4])
            for t in wl:
                fills.append(t[1] - t[0]); loops.append(t[2] - t[1]); epis.append(t[3] - t[2]); drains.append(t[4] - t[3])
            share.append(sum(t[2] - t[1] for t in wl) / (wl[-1][4] - wl[0][0]))
        print(f"{name} {op} ({kern}): {nblk} workgroups on {len(per_cu)} CUs, {min(len(v) for v in per_cu.values())}-"
              f"{max(len(v) for v in per_cu.values())} per CU")
        for lab, v in (("gap end -> next start", gaps), ("fill", fills), ("loop", loops), ("epilogue", epis), ("store drain", drains)):
            print(f"    {lab:22s} median {pct(v, 0.5):7d}  p10 {pct(v, 0.1):7d}  p90 {pct(v, 0.9):7d} cycles")
        print(f"    K-loop share of a CU's span, first start to last end: median {pct(share, 0.5):.3f}  min {min(share):.3f}", flush=True)
