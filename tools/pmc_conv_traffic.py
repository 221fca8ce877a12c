#!/usr/bin/env python3
"""HBM-side traffic of the conv kernel set per training step from two rocprofv3 --pmc passes (tools/pmc_conv_traffic.sh):
    python tools/pmc_conv_traffic.py gpurun_out/ct_<tag> <steps>  > profiles/rNN/pmc_traffic_conv_step.txt
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a
16-B-per-lane read (buffer_load ... lds and global_load alike) at half its bytes -> doubled; WRITE_SIZE is exact for 16-B-per-
lane stores.  Every load and store of these kernels is 16 B per lane except the 2-bit pool records (8 B per lane) and
the bias-gradient rows (4 B), which are < 1 % of the bytes.  Infinity-Cache hits are counted (memory-side requests of the L2)."""
import collections
import csv
import json
import re
import sys

root, steps = sys.argv[1], int(sys.argv[2])
CONV = re.compile(r"igemm8?_kernel|wgrad8?_kernel|stem_pool_kernel|splitk_reduce_kernel|splitk_epilogue_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    path = f"{root}/pmc_{c}/step_counter_collection.csv"
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == c:
            k = short(r["Kernel_Name"])
            if CONV.search(k) and "false, true>" not in k:       # (not fc.1's fused weight gradient + SGD)
                agg[k][0] += float(r["Counter_Value"])
                agg[k][1] += 1
    val[c] = agg

print("HBM-side traffic of the conv kernel set per training step (batch 64), rocprofv3 --pmc, separate passes")
print(f"command: rocprofv3 --pmc <counter> --output-format csv -- python3 tools/step_trace.py {steps}   (tools/pmc_conv_traffic.sh)")
print("Counter unit: KB per dispatch, summed over the dispatches of a kernel and divided by the steps of the run.")
print("gfx950 correction: FETCH_SIZE doubled (16 B/lane reads are tallied at half their bytes), WRITE_SIZE exact.\n")
print(f"{'kernel':48s} {'calls/step':>10s} {'read MB':>9s} {'write MB':>9s} {'MB/step':>9s}")
tr = tw = 0.0
for k in sorted(val["FETCH_SIZE"], key=lambda k: -(2 * val["FETCH_SIZE"][k][0] + val["WRITE_SIZE"].get(k, [0, 0])[0])):
    f, n = val["FETCH_SIZE"][k]
    w = val["WRITE_SIZE"].get(k, [0.0, 0])[0]
    rd, wr = 2 * f / 1e3 / steps, w / 1e3 / steps
    tr += rd
    tw += wr
    print(f"{k:48s} {n / steps:10.1f} {rd:9.1f} {wr:9.1f} {rd + wr:9.1f}")
print(f"\nconv set per step: read {tr / 1e3:.2f} GB + write {tw / 1e3:.2f} GB = {(tr + tw) / 1e3:.2f} GB")
print("#json " + json.dumps({"conv": round((tr + tw) * 1e6)}))     # bytes per step; read by bench.py
