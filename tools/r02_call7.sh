#!/bin/bash
# round-2 GPU call 7: profile of the current state (which tiles the autotune picks in situ) + stream ablations
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_7
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export PXL_STATS_REP=4 PXL_FUSE_BN_FINALIZE=1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
db=$(find $OUT/prof -name "*results.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" $OUT/kernel_stats.csv > /dev/null && python tools/prof_summary.py --one-step "$db" $OUT/step_breakdown.txt | head -45
cp "$db" $OUT/trace.db 2>/dev/null; ls -la $OUT/trace.db
rm -rf $OUT/prof
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events"
PXL_TEACHER_STREAM=0 timeout 300 $B > $OUT/b_noteacherstream.json 2> $OUT/b.err
PXL_SIDE_STREAM=0 timeout 300 $B > $OUT/b_nowgradstream.json 2>> $OUT/b.err
timeout 300 python bench.py --algo suponly --lbs 8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/b_suponly8.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
