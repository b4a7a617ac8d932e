#!/bin/bash
# bisect of the AdvSSL / GCT slowdown over the commits of this round (worktrees under tmp_bisect/, each with its own build)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
OUT=$ROOT/gpurun_out/r02_22
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 59b8774 d369ddb 85d37bb 7241a4d 302dcc9; do
  cd $ROOT/tmp_bisect/$c
  for a in adv gct; do
    timeout 300 python bench.py --algo $a --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou > $OUT/b_${c}_$a.json 2>> $OUT/b.err || \
    timeout 300 python bench.py --algo $a --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/b_${c}_$a.json 2>> $OUT/b.err
  done
done
cd $ROOT
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
tail -3 $OUT/b.err
