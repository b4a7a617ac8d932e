#!/bin/bash
# round-2 GPU call 6: tall conv_dma tiles -- kernel tests, per-shape micro-benchmark, step bench with the extended autotune
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_6
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "conv_dma or bnreduce or bn_backward_reduce" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/kern.log
timeout 600 python tools/conv_bench.py --dtype bf16 --modes fwd,dgrad --cfgs=-1,8,10,20,21,22,23,24,26 --only l3 > $OUT/cb_l3.txt 2>&1
timeout 600 python tools/conv_bench.py --dtype bf16 --modes fwd,dgrad --cfgs=-1,8,10,20,21,22,23,24,26 --only l4 > $OUT/cb_l4.txt 2>&1
cat $OUT/cb_l3.txt $OUT/cb_l4.txt | grep -v amdgpu
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events"
timeout 300 $B > $OUT/b_default.json 2> $OUT/b.err
PXL_STATS_REP=4 PXL_FUSE_BN_FINALIZE=1 timeout 300 $B > $OUT/b_rep4_fuse.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_losses"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 $OUT/b.err
