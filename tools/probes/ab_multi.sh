#!/bin/bash
# Same-box interleaved comparison of bench.py under several environments:
#   ab_multi.sh ROUNDS "A=1 B=2" "A=0" ...     (each argument after ROUNDS is one configuration, "-" = plain)
set -o pipefail
rounds=$1; shift
mkdir -p gpurun_out/r3/abm
n=0
for cfg in "$@"; do n=$((n + 1)); done
for r in $(seq 1 $rounds); do
    c=0
    for cfg in "$@"; do
        c=$((c + 1))
        if [ "$cfg" = "-" ]; then cfg=""; fi
        env $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-nms --no-cpu-baseline > gpurun_out/r3/abm/c${c}_$r.json 2> gpurun_out/r3/abm/c${c}_$r.err || exit 1
    done
done
python - "$rounds" "$@" <<'PY'
import json, sys
rounds = int(sys.argv[1])
for c, cfg in enumerate(sys.argv[2:], 1):
    rs = [json.load(open(f"gpurun_out/r3/abm/c{c}_{r}.json")) for r in range(1, rounds + 1)]
    print(f"{cfg:40s}", [r["ms_per_step"] for r in rs], "loss", rs[0]["final_loss"])
PY
