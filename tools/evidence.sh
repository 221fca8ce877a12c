#!/bin/bash
# Evidence pass on the GPU box (run through gpurun from the repo root):  bash tools/evidence.sh <tag>
#   1. bench.py as the driver runs it                       -> gpurun_out/ev_<tag>/bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> gpurun_out/ev_<tag>/prof/*_kernel_stats.csv
#   3. FETCH_SIZE / WRITE_SIZE of the three HBM-bound kernel sets, one --pmc pass each
# Every GPU step has its own time limit; the script stops at the first failure.
set -eo pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ev_$tag
mkdir -p "$out"
cd "$root"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
echo "bench done" >&2
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -o bench --output-format csv -- \
    python3 "$root/bench.py" --steps 20 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
echo "trace done" >&2
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c -d "$out/pmc_$c" -o hbm --output-format csv -- \
        python3 "$root/tools/hbm_kernels_bench.py" > "$out/pmc_$c.log" 2>&1
    echo "pmc $c done" >&2
done
# keep what travels back small: the per-dispatch trace of the bench is not needed, the stats are
find "$out/prof" -name '*kernel_trace.csv' -delete
ls -R "$out" | head -50
