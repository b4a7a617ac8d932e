#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_18
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "stem or fused_task or cross_entropy" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/kern.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2>> $OUT/b.err; }
run all X=1
run nostem PXL_STEM_PATCHES=0
run all2 X=1
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
tail -3 $OUT/b.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-miou > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof -name "*results.db" | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py "$DB" $OUT/kernel_stats.csv "r02_18" > /dev/null; python tools/prof_summary.py --one-step "$DB" $OUT/step_breakdown.txt > /dev/null; cp "$DB" $OUT/trace.db; fi
rm -rf $OUT/prof
head -6 $OUT/step_breakdown.txt
