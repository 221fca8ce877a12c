#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: ms per step and calls per step per kernel (names shortened).
    python tools/prof_summary.py <kernel_stats.csv> <steps incl. warmup> [top]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time {tot / 1e6 / steps:.3f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:90]
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls/step  avg {float(r['AverageNs']) / 1e3:9.1f} us  {name}")
