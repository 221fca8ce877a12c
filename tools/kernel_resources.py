#!/usr/bin/env python3
"""Compiler resource usage of every conv kernel instantiation -> profiles/rNN/kernel_resources.txt
    python tools/kernel_resources.py > profiles/r02/kernel_resources.txt
Runs hipcc -Rpass-analysis=kernel-resource-usage on csrc/conv_kernels.hip (device pass only) and tabulates it."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "yolo_from_scratch_amd", "csrc", "conv_kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(root, "include"), "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# Compiler resource usage of every conv kernel instantiation (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage,")
print("# ROCm 7.2, device pass; tools/kernel_resources.py).  LDS is dynamic (extern __shared__): igemm8 131072 + 16384 B, wgrad8 131072 B,")
print("# stem_pool_kernel 2 x 7 input rows (102400 B at 448^2) - one workgroup of 512 threads per CU each; igemm_kernel")
print("# STAGES*(BM+BN)*128 B or its epilogue slab, wgrad_kernel 2*(BX+BZ)*128 B.  waves/SIMD is the compiler's register-limited occupancy.")
print("# template arguments: igemm_kernel<BM,BN,WM,WN,STAGES,EPI,SUB> (EPI 0 bias+leaky, 1 mask, 2 none, 3 pool), igemm8_kernel<EPI,FP1>,")
print("# wgrad_kernel<BX,BZ,POOLZ,SGD>, wgrad8_kernel<FX>, stem_pool_kernel<CB,LMAX>.")
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s} {'VGPR spill':>11s} {'SGPR spill':>11s}")
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0]
    if not re.match(r"(igemm|wgrad|stem_pool)", n):
        continue
    print(f"{n:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('ScratchSize [bytes/lane]', '?'):>15s} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>11s} {r.get('VGPRs Spill', '?'):>11s} {r.get('SGPRs Spill', '?'):>11s}")
