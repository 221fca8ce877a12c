#!/usr/bin/env python3
"""Timeline of the train step from a rocprofv3 --kernel-trace CSV: how much of a step's wall time has 0, 1 or >= 2
kernels on the GPU, and which kernels own the time in which they run alone.
    python tools/timeline.py <..._kernel_trace.csv> [steps_to_skip]
Steps are delimited by the last sgd_kernel launch of each step (the optimiser ends the step)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")) for r in rows))
# step boundaries: end of the last kernel before a gap-free run restarts with the stem's space-to-depth kernel
bounds = [s for s, e, n, q in ev if n.startswith("stem_s2d_kernel") or "stem_s2d" in n]
if len(bounds) < skip + 2:
    raise SystemExit("not enough steps in the trace")
t0, t1 = bounds[skip], bounds[-1]
nsteps = len(bounds) - 1 - skip
sel = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in ev if e > t0 and s < t1]
pts = []
for i, (s, e, n, q) in enumerate(sel):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active = set()
last = t0
hist = collections.Counter()
alone = collections.Counter()
pairs = collections.Counter()
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        k = len(active)
        hist[min(k, 2)] += dt
        if k == 1:
            alone[sel[next(iter(active))][2]] += dt
        elif k >= 2:
            names = sorted(sel[j][2] for j in active)[:2]
            pairs[" + ".join(names)] += dt
    last = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
wall = (t1 - t0) / nsteps / 1e6
print(f"{nsteps} steps, {wall:.3f} ms per step (first kernel of a step to the first kernel of the next)")
for k, label in ((0, "no kernel on the GPU"), (1, "exactly one kernel"), (2, "two or more kernels")):
    print(f"  {label:24s} {hist[k] / nsteps / 1e6:7.3f} ms/step  {100.0 * hist[k] / (t1 - t0):5.1f} %")
print("time alone on the GPU, by kernel (ms/step):")
for n, v in alone.most_common(14):
    print(f"  {v / nsteps / 1e6:7.3f}  {n}")
print("overlapped time, by pair (ms/step):")
for n, v in pairs.most_common(8):
    print(f"  {v / nsteps / 1e6:7.3f}  {n}")
busy = collections.Counter()
for s, e, n, q in sel:
    busy[n] += e - s
print("kernel duration sums (ms/step):")
for n, v in busy.most_common(12):
    print(f"  {v / nsteps / 1e6:7.3f}  {n}")
