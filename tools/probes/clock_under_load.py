#!/usr/bin/env python3
"""Shader clock and power while one kernel family runs back to back (rocm-smi sampled from a thread every ~0.5 s):
is the 2.4 GHz the dense-MFMA peak is quoted at reachable under the conv kernels' load?
    python tools/probes/clock_under_load.py [seconds per load]
Loads: idle, register-only MFMA loop (yfs_probe_mfma_bf16), feat.18 forward (igemm8_kernel), feat.18 weight gradient
(wgrad8_kernel), a device-to-device copy (HBM only); the vendor GEMM (torch.mm -> hipBLASLt, 8192^3) on random and on zero operands; the forward also on an all-zero input (same instructions, no toggling
operands: what the data itself costs in power)."""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import conv_ops as co, _lib as L
import ctypes

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda")
x = co.nhwc(torch.rand(64, 512, 28, 28, device=dev) - 0.3, torch.bfloat16)
w = co.nhwc((torch.rand(1024, 512, 3, 3, device=dev) * 2 - 1) / 68.0)
b = torch.zeros(1024, device=dev)
w16, _ = co.weight_prep(w, want_wt16=False)
dz = co.nhwc(torch.randn(64, 1024, 28, 28, device=dev), torch.bfloat16)
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
big2 = torch.empty_like(big)
scratch = torch.zeros(16, device=dev)
flop = ctypes.c_double(0)


def mfma():
    L.check(L.lib().yfs_probe_mfma_bf16(L.c_int(20000), L.c_int(1024), ctypes.byref(flop), L.ptr(scratch), L.stream()), "probe")


def mfma_ex(shape, toggling):
    def run():
        L.check(L.lib().yfs_probe_mfma_bf16_ex(L.c_int(shape), L.c_int(toggling), L.c_int(20000), L.c_int(1024), ctypes.byref(flop),
                                               L.ptr(scratch), L.stream()), "probe_ex")
    return run


x0 = torch.zeros_like(x)
ga = (torch.rand(8192, 8192, device=dev) - 0.5).to(torch.bfloat16)
gb = (torch.rand(8192, 8192, device=dev) - 0.5).to(torch.bfloat16)
gz = torch.zeros_like(ga)
gout = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
MFMA_FLOP = 1024 * 4.0 * 20000 * 16.0 * (2.0 * 16 * 16 * 32)
L18_FLOP = 2.0 * 64 * 28 * 28 * 1024 * 9 * 512
FLOPS = {"mfma": MFMA_FLOP, "feat.18": L18_FLOP, "hipBLASLt": 2.0 * 8192 ** 3}
loads = [("idle", None), ("mfma register loop", mfma),
         ("mfma 16x16x32, operands = 1", mfma_ex(0, 0)), ("mfma 16x16x32, random operands", mfma_ex(0, 1)),
         ("mfma 32x32x16, operands = 1", mfma_ex(1, 0)), ("mfma 32x32x16, random operands", mfma_ex(1, 1)), ("feat.18 forward (igemm8)", lambda: co.conv_fwd(x, w16, b, 1)),
         ("feat.18 forward, x = 0", lambda: co.conv_fwd(x0, w16, b, 1)),
         ("feat.18 wgrad (wgrad8)", lambda: co.conv_wgrad(x, dz, 3, 1)), ("hipBLASLt 8192^3 bf16, random", lambda: torch.mm(ga, gb, out=gout)),
         ("hipBLASLt 8192^3 bf16, zeros", lambda: torch.mm(gz, gz, out=gout)),
         ("d2d copy 256 MiB", lambda: big2.copy_(big))]


def sample(out, stop):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            out.append(json.loads(r.stdout))
        except Exception as e:      # noqa: BLE001
            out.append({"error": str(e)})
        time.sleep(0.3)


for name, fn in loads:
    out, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(out, stop))
    t0 = time.time()
    th.start()
    n = 0
    while time.time() - t0 < secs:
        if fn is None:
            time.sleep(0.1)
            continue
        for _ in range(20):
            fn()
            n += 1
        torch.cuda.synchronize()
    dt = time.time() - t0
    stop.set()
    th.join()
    sclk, power = [], []
    for s in out[1:]:                      # the first sample may predate the load
        for card, d in s.items():
            if not isinstance(d, dict):
                continue
            for k, v in d.items():
                if "sclk" in k.lower() and "(" in str(v):
                    sclk.append(float(str(v).split("(")[1].split("M")[0]))
                if "power" in k.lower() and "(w)" in k.lower():
                    try:
                        power.append(float(v))
                    except ValueError:
                        pass
    fmt = lambda a: f"{min(a):.0f}-{max(a):.0f} (mean {sum(a) / len(a):.0f})" if a else "n/a"
    fl = next((v for k, v in FLOPS.items() if name.startswith(k)), 0.0)
    print(f"{name:32s} {n / dt if fn else 0:9.1f} launches/s {fl * n / dt / 1e12 if fn else 0:7.0f} TFLOP/s  sclk MHz {fmt(sclk)}  power W {fmt(power)}  [{len(out)} samples]", flush=True)
if out:
    print("last raw sample:", json.dumps(out[-1])[:600])
