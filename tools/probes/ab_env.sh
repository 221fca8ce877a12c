#!/bin/bash
# Same-box interleaved A/B of bench.py with an environment variable off / on:  ab_env.sh VAR OFFVALUE ONVALUE [pairs]
# Prints ms_per_step and the conv roofline fraction of every run.
set -o pipefail
var=$1; off=$2; on=$3; pairs=${4:-3}
mkdir -p gpurun_out/r3
for i in $(seq 1 $pairs); do
    env $var=$off timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-nms --no-cpu-baseline > gpurun_out/r3/ab0_$i.json 2> gpurun_out/r3/ab0_$i.err || exit 1
    env $var=$on timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-nms --no-cpu-baseline > gpurun_out/r3/ab1_$i.json 2> gpurun_out/r3/ab1_$i.err || exit 1
done
python - "$pairs" <<'PY'
import json, sys
n = int(sys.argv[1])
for k in ("ab0", "ab1"):
    rs = [json.load(open(f"gpurun_out/r3/{k}_{i}.json")) for i in range(1, n + 1)]
    print(k, [(r["ms_per_step"], r["roofline"]["frac"], r["final_loss"]) for r in rs])
PY
