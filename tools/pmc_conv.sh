#!/bin/bash
# PMC passes over tools/conv_bench.py (separate rocprofv3 runs per counter group; --kernel-trace only).
# (the TA_* counter pass hung on this image in round 3 and is no longer run)
# usage: tools/pmc_conv.sh <conv_bench args ...>   -> gpurun_out/pmc/pass*.csv + summary
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---only l3.3x3 --cfgs 9,10 --modes fwd --iters 2}"
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/tools/conv_bench.py $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
