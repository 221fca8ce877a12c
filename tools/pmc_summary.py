#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over dispatches."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
