#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_15
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "join or fused_task or dgrad_with_fused or wgrad_dma or cross_entropy" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/kern.log
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_multistep.py tests/test_checkpoint.py tests/test_psp.py -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/net.log 2>&1; echo "net rc=$?"; tail -5 $OUT/net.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2>> $OUT/b.err; }
run all X=1
run nojoin PXL_FUSE_JOIN=0
run fork1 PXL_FORK_EVERY=1
run fork6 PXL_FORK_EVERY=6
run noloss PXL_FUSE_MT_LOSS=0
run tprio PXL_TEACHER_PRIO=-1
run all2 X=1
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
# one traced run (kernel-trace only) for the step breakdown
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-miou > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof -name "*results.db" | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py "$DB" $OUT/kernel_stats.csv "r02_15" > /dev/null; python tools/prof_summary.py --one-step "$DB" $OUT/step_breakdown.txt > /dev/null; cp "$DB" $OUT/trace.db; fi
rm -rf $OUT/prof
head -5 $OUT/step_breakdown.txt
# HBM-side traffic of the contraction kernels: two PMC passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$c -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-miou > $OLDPWD/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
