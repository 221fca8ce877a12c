#!/usr/bin/env python3
"""igemm8_kernel persistent (one workgroup per CU walks its tiles) against one workgroup per tile (YFS_KNOB_IGEMM_PERSIST =
1) while another queue holds CUs for milliseconds, as RCCL's kernels do in a data-parallel step: `hog` workgroups of the
register-only MFMA probe run on a second stream (a CU that hosts one cannot host an igemm8 workgroup: that needs the
whole register file), feat.7's forward (1792 tiles) and feat.18's masked dgrad (448 tiles) are timed beside them."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from yolo_from_scratch_amd import _lib, conv_ops as co
lib = _lib.lib()
co.SPLIT_K = False
side = torch.cuda.Stream()
scratch = torch.zeros(1 << 20, device="cuda")
flop = ctypes.c_double()


def hog(blocks, iters):
    if blocks:
        with torch.cuda.stream(side):
            _lib.check(lib.yfs_probe_mfma_bf16(ctypes.c_int(iters), ctypes.c_int(blocks), ctypes.byref(flop), _lib.ptr(scratch),
                                               ctypes.c_void_p(side.cuda_stream)), "probe")


def timed(fn, blocks, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    hog(blocks, 400000)               # several ms: outlasts the timed launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(200000)         # let the hog settle on its CUs first
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x7 = co.nhwc((torch.rand(64, 256, 56, 56, device="cuda") - 0.3), torch.bfloat16)
w7 = co.weight_prep(co.nhwc((torch.rand(512, 256, 3, 3, device="cuda") * 2 - 1) / 48), want_wt16=False)[0]
b7 = torch.zeros(512, device="cuda")
dz18 = co.nhwc(torch.randn(64, 1024, 28, 28, device="cuda"), torch.bfloat16)
a18 = co.nhwc(torch.randn(64, 512, 28, 28, device="cuda"), torch.bfloat16)
wt18 = co.weight_prep(co.nhwc((torch.rand(1024, 512, 3, 3, device="cuda") * 2 - 1) / 68))[1]
cases = (("feat.7 forward", lambda: co.conv_fwd(x7, w7, b7, 1)), ("feat.18 masked dgrad", lambda: co.conv_dgrad(dz18, wt18, (28, 28), 1, x_act=a18)))
for name, fn in cases:
    for blocks in (0, 16, 32, 64):
        row = []
        for knob in (0, 1):
            _lib.set_knob("igemm_persist", knob)
            row.append(timed(fn, blocks))
        _lib.set_knob("igemm_persist", 0)
        print(f"{name:22s} hog workgroups {blocks:3d}: persistent {row[0]:7.1f} us   one workgroup per tile {row[1]:7.1f} us", flush=True)
