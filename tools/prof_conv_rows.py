#!/usr/bin/env python3
"""Conv rows of a rocprofv3 *_kernel_stats.csv against bench.py's roofline.conv_ms_per_step of the same run.
    python tools/prof_conv_rows.py <kernel_stats.csv> <bench.json>
The process runs warm-up + timed + 1 + roofline-pass steps (bench.py), all with the weight-gradient stream off."""
import csv
import json
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rec = json.loads([ln for ln in open(sys.argv[2]) if ln.startswith("{")][-1])
rsteps = max(3, min(10, rec["steps"]))
steps = rec["warmup"] + rec["steps"] + 1 + rsteps
CONV = re.compile(r"igemm8?_kernel|wgrad8?_kernel|stem_pool_kernel|splitk_reduce_kernel|splitk_epilogue_kernel")
print(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {rec['steps']} --warmup {rec['warmup']} --no-nms "
      f"--no-cpu-baseline, YFS_WGRAD_STREAM=0 ({steps} steps in the process)")
print(f"# bench.py of this run: {rec['ms_per_step']} ms/step (single stream), roofline.conv_ms_per_step "
      f"{rec['roofline']['conv_ms_per_step']}, frac {rec['roofline']['frac']}")
tot = 0.0
out = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    if not CONV.search(name) or "sgd" in name.lower() and "true>" in name.replace(" ", "")[-6:]:
        continue
    if re.search(r"wgrad_kernel<128, 128, false, true>", name):     # fc.1's fused weight gradient + SGD: not a conv launch
        continue
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    out.append((ms, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, name))
for ms, calls, avg, name in sorted(out, reverse=True):
    print(f"{ms:8.3f} ms/step {calls:6.1f} calls/step  avg {avg:9.1f} us  {name[:100]}")
conv = rec["roofline"]["conv_ms_per_step"]
print(f"sum of the conv rows {tot:.3f} ms/step; bench.py brackets {conv:.3f} ms/step; ratio {tot / conv:.4f}")
