#!/bin/bash
# round-2 GPU call 1: baseline bench, BN-finalize fusion / statistics-replica experiments, step profile, fp32 backward diagnostic
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_1
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events"
timeout 300 $B > $OUT/b_default.json 2> $OUT/b_default.err
PXL_STATS_REP=4 timeout 300 $B > $OUT/b_rep4.json 2>> $OUT/b_default.err
PXL_STATS_REP=4 PXL_FUSE_BN_FINALIZE=1 timeout 300 $B > $OUT/b_rep4_fuse.json 2>> $OUT/b_default.err
PXL_STATS_REP=1 PXL_FUSE_BN_FINALIZE=1 timeout 300 $B > $OUT/b_rep1_fuse.json 2>> $OUT/b_default.err
PXL_STATS_REP=8 PXL_FUSE_BN_FINALIZE=1 timeout 300 $B > $OUT/b_rep8_fuse.json 2>> $OUT/b_default.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
timeout 300 python tools/diag_fp32_grad.py > $OUT/diag_fp32.txt 2>&1; echo "diag rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
db=$(find $OUT/prof -name "*results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" $OUT/kernel_stats.csv > /dev/null && python tools/prof_summary.py --one-step "$db" $OUT/step_breakdown.txt | head -5
rm -rf $OUT/prof
grep -E "==|more than" $OUT/diag_fp32.txt
