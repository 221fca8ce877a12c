#!/bin/bash
# conv PMC evidence: run on the GPU box from the repo root
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/${PMC_TAG:-pmc}
mkdir -p $OUT
CMD="python3 $R/tools/conv_bench.py --iters 3 --ops step --layers 0,2,5,7,9,18,24,26"
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pass $i: $ctrs"
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace -d $OUT/p$i -o pmc -f csv -- $CMD > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
find $OUT -name "*.csv" | head -20
