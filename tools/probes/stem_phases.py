#!/usr/bin/env python3
"""Section clocks of stem_pool_kernel (needs libyfs_stemdbg.so: `make -C yolo_from_scratch_amd/csrc stem-timing`; STEM_T marks in
conv_kernels.hip).  Run as  YFS_LIB_PATH=yolo_from_scratch_amd/libyfs_stemdbg.so python tools/probes/stem_phases.py
Prints, for the 8 waves of workgroup 7 of a batch-64 448^2 launch, cycles per tile in the MFMA section and in the epilogue,
and cycles per strip waiting at the strip barrier."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()
x = torch.rand(64, 3, 448, 448, device="cuda") - 0.4
w = (torch.randn(192, 3, 7, 7, device="cuda") * 0.05)
b = torch.randn(192, device="cuda") * 0.2
xs = co.stem_space_to_depth(x)
w2 = co.stem_weight(co.nhwc(w))
oh, ow, _, _ = co.stem_geometry(448, 448)
for _ in range(3):
    co.stem_fwd_pool(xs, w2, b, oh, ow)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
assert lib.yfs_debug_stem_times(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); co.stem_fwd_pool(xs, w2, b, oh, ow); e1.record()
torch.cuda.synchronize()
assert lib.yfs_debug_stem_times(buf, 0) == 0
print("launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
for wv in range(8):
    m, e, n, bar = (buf[wv * 4 + k] for k in range(4))
    print(f"wave {wv}: tiles {n}  mfma section {m / max(n, 1):.0f} cyc/tile  epilogue {e / max(n, 1):.0f} cyc/tile  barrier wait {bar / max(n / 7, 1):.0f} cyc/strip")
