#!/bin/bash
# HBM-side traffic of the conv kernel set over whole training steps (run through gpurun from the repo root):
#   bash tools/pmc_conv_traffic.sh <tag>   -> gpurun_out/ct_<tag>/pmc_{FETCH,WRITE}_SIZE/...
# Two separate --pmc passes (FETCH_SIZE alone, WRITE_SIZE alone; MI355X_MICROARCH.md, HBM section) over
# tools/step_trace.py (the bench's train step, batch 64), the program directly after "--", no tracing domains.
set -eo pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ct_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c -d "$out/pmc_$c" -o step --output-format csv -- \
        python3 "$root/tools/step_trace.py" 4 > "$out/pmc_$c.log" 2>&1
    echo "pmc $c done" >&2
done
python3 "$root/tools/pmc_conv_traffic.py" "$out" 4
