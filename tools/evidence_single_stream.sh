#!/bin/bash
# Evidence that reproduces bench.py's headline roofline (run through gpurun from the repo root):
#   bash tools/evidence_single_stream.sh <tag>
# bench.py's roofline.achieved divides the conv FLOPs of a step by HIP-event brackets taken with the weight-gradient
# stream off.  This runs the same command with that stream off for the WHOLE process under rocprofv3 --kernel-trace
# --stats, (and the overlapped optimiser tail), so the per-kernel average durations in the stats are those of kernels running alone; the conv rows (igemm*,
# wgrad*, stem_pool, splitk_*) summed per step must agree with roofline.conv_ms_per_step of the same run.
set -eo pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ss_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
export YFS_WGRAD_STREAM=0
export YFS_OVERLAP_TAIL=0     # nor the optimiser tail of the previous step behind the first convolutions
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -o bench --output-format csv -- \
    python3 "$root/bench.py" --steps 20 --warmup 5 --no-nms --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
find "$out/prof" -name '*kernel_trace.csv' -delete
stats=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
python3 "$root/tools/prof_conv_rows.py" "$stats" "$out/bench.json" > "$out/bench_kernel_summary_single_stream.txt"
cat "$out/bench_kernel_summary_single_stream.txt"
