#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_conv.sh) per conv kernel: python tools/pmc_summarize.py gpurun_out/r2/pmc"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values per dispatch
dur = defaultdict(list)
res = {}
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(igemm8?_kernel<[^>]*>|wgrad8?_kernel<[^>]*>|stem_pool_kernel<[^>]*>|splitk_reduce_kernel<[^>]*>|splitk_epilogue_kernel)", name)
        if not m:
            continue
        k = m.group(1).replace(" ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        res[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size"])
CUS, SIMDS, XCDS = 256, 4, 8
print("# per dispatch averages; tools/conv_bench.py --ops step (the launches of the training step, batch 64) --layers 0,2,5,7,9,18,24,26; one counter group per rocprofv3 run")
# which launches of the step run on each kernel instantiation (layers of this run)
LAYER = {"igemm8_kernel<3,4,2>": "feat.2 forward + pool", "igemm8_kernel<3,3,2>": "feat.7 / feat.18 forward + pool",
         "igemm8_kernel<0,3,2>": "feat.5 / feat.24 forward", "igemm8_kernel<1,3,2>": "feat.7 / 18 / 24 masked dgrad",
         "igemm8_kernel<1,4,1>": "feat.2 masked dgrad (192 output channels)", "igemm8_kernel<4,3,2>": "feat.26 forward and dgrad, split-K",
         "wgrad8_kernel<4>": "feat.7 / 18 / 24 / 26 weight gradient", "wgrad8_kernel<3>": "feat.2 weight gradient",
         "stem_pool_kernel<3,true>": "feat.0 forward + pool", "wgrad_kernel<128,192,true,false>": "feat.0 weight gradient from the pooled side",
         "wgrad_kernel<128,128,false,false>": "feat.5 / feat.9 weight gradient", "igemm_kernel<128,128,2,2,2,0,false>": "feat.9 forward (1x1)",
         "igemm_kernel<64,128,2,2,2,1,false>": "feat.9 masked dgrad (1x1)", "igemm_kernel<128,128,2,2,2,1,false>": "feat.5 masked dgrad"}
import json
summary = {}
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 4 SIMDs x 256 CUs); clock = GRBM_GUI_ACTIVE / 8 / duration")
rows = sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]]))
for k, c in rows:
    n = len(dur[k])
    avg = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else None   # noqa: E731
    gui = avg("GRBM_GUI_ACTIVE")
    d = sum(dur[k]) / n
    line = [f"{k:48s} n={n:3d} avg {d:8.1f} us"]
    mf = avg("SQ_VALU_MFMA_BUSY_CYCLES")
    if mf and gui:
        line.append(f"mfma_busy {mf / (gui / XCDS * SIMDS * CUS):.3f}")
        summary[k] = {"mfma_busy": round(mf / (gui / XCDS * SIMDS * CUS), 3), "avg_us": round(d, 1), "launches": LAYER.get(k, "")}
        line.append(f"clock {gui / XCDS / (d * 1e3) :.2f} GHz")
    wc, wa, wi, ai = avg("SQ_WAVE_CYCLES"), avg("SQ_WAIT_ANY"), avg("SQ_WAIT_INST_ANY"), avg("SQ_ACTIVE_INST_ANY")
    if wc:
        line.append(f"wait_any/wave_cycles {wa / wc:.2f} wait_inst {wi / wc:.2f} active_inst {ai / wc:.2f}")
    bc, la = avg("SQ_LDS_BANK_CONFLICT"), avg("SQ_LDS_IDX_ACTIVE")
    if la:
        line.append(f"lds_bank_conflict/lds_active {bc / la:.4f}")
        if gui:
            line.append(f"lds_busy {la / (gui / XCDS * CUS):.2f}")
    iv, im, il = avg("SQ_INSTS_VALU"), avg("SQ_INSTS_MFMA"), avg("SQ_INSTS_LDS")
    if im:
        line.append(f"valu/mfma {iv / im:.2f} lds/mfma {il / im:.2f}")
    h, mi = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
    if h is not None and mi is not None and h + mi > 0:
        line.append(f"l2_hit {h / (h + mi):.3f}")
    ta = avg("TA_TA_BUSY_sum")
    if ta and gui:
        line.append(f"ta_busy {ta / (gui / XCDS * CUS):.2f}")
    rd, wr = avg("TCC_EA0_RDREQ_sum"), avg("TCC_EA0_WRREQ_sum")
    if rd is not None and wr is not None:
        line.append(f"ea_rdreq {rd:.3g} ea_wrreq {wr:.3g}")
    v = res[k]
    line.append(f"| vgpr {v[0]} agpr {v[1]} sgpr {v[2]} lds {v[3]} scratch {v[4]} wg {v[5]}")
    if k in LAYER:
        line.append(f"| {LAYER[k]}")
    print("  ".join(line))
print("#json " + json.dumps(summary))     # read by bench.py (roofline.mfma_busy_pmc)
