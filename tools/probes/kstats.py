"""Per-kernel average durations from a rocprofv3 --kernel-trace csv (names contain commas: use the csv module)."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{len(v):5d} calls  median {v2[len(v2)//2]:8.1f} us  min {v2[0]:8.1f}  {k[:110]}")
