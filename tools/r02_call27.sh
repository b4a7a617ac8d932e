#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_27
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="python -m pytest tests/test_gpu_dist.py -m gpu -q --tb=line -p no:cacheprovider -k two_ranks_one_gpu -s"
for v in "X=1" "X=2" "PXL_SIDE_STREAM=0" "PXL_PACK_STREAM=0" "PXL_GRAD_OVERLAP=0" "PXL_SIDE_PRIO=1"; do
  echo "=== $v" >> $OUT/dist.log
  env $v timeout 300 $T 2>&1 | grep -E "2-rank vs full|passed|failed|worst" | cut -c1-400 >> $OUT/dist.log
done
cat $OUT/dist.log
