#!/bin/bash
# round-2 GPU call 8: stride-2 dgrads on the DMA kernel + threaded teacher enqueue -- tests, micro-benchmark, bench, profile
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_8
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "conv_dma or bnreduce" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/kern.log
timeout 900 python -m pytest tests/test_multistep.py -m gpu -q -x --tb=short -p no:cacheprovider -k "mt_six or suponly_six" > $OUT/ms.log 2>&1; echo "multistep rc=$?"; tail -3 $OUT/ms.log
timeout 600 python tools/conv_bench.py --dtype bf16 --modes dgrad --cfgs=-1,0,8,10,24,26 --only s2 > $OUT/cb_s2.txt 2>&1
timeout 600 python tools/conv_bench.py --dtype bf16 --modes dgrad --cfgs=-1,0,8,10,24,26 --only ds > $OUT/cb_ds.txt 2>&1
cat $OUT/cb_s2.txt $OUT/cb_ds.txt | grep -v amdgpu
export PXL_STATS_REP=4 PXL_FUSE_BN_FINALIZE=1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events"
timeout 300 $B > $OUT/b_thread.json 2> $OUT/b.err
PXL_ENQUEUE_THREAD=0 timeout 300 $B > $OUT/b_nothread.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_losses"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
db=$(find $OUT/prof -name "*results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py --one-step "$db" $OUT/step_breakdown.txt | head -6
cp "$db" $OUT/trace.db 2>/dev/null
rm -rf $OUT/prof
tail -3 $OUT/b.err
