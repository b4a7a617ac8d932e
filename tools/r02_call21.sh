#!/bin/bash
# why are AdvSSL / GCT / CCT slower than in round 1?  A/B switches + one kernel-stats run each
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_21
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
run() { name=$1; algo=$2; shift; shift; env "$@" timeout 300 python bench.py --algo $algo $Q > $OUT/b_${algo}_$name.json 2>> $OUT/b.err; }
run default adv X=1
run s2dma16 adv PXL_DMA_STRIDED_DGRAD=16
run nojoin adv PXL_FUSE_JOIN=0
run fork3 adv PXL_FORK_EVERY=3
run nofin adv PXL_FUSE_BN_FINALIZE=0
run nostreams adv PXL_ADV_STREAMS=0
run default cct X=1
run nofin cct PXL_FUSE_BN_FINALIZE=0
run rep32 cct PXL_STATS_REP=32
run nojoin cct PXL_FUSE_JOIN=0
run default gct X=1
run s2dma16 gct PXL_DMA_STRIDED_DGRAD=16
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
for a in adv cct; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_$a -o p -- python $OLDPWD/bench.py --algo $a --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-miou > $OLDPWD/$OUT/prof_$a.log 2>&1); echo "prof $a rc=$?"
  DB=$(find $OUT/prof_$a -name "*results.db" | head -1)
  [ -n "$DB" ] && python tools/prof_summary.py "$DB" $OUT/kernel_stats_$a.csv "$a" | head -32 > $OUT/top_$a.txt
  rm -rf $OUT/prof_$a
done
tail -3 $OUT/b.err
