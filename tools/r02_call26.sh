#!/bin/bash
# final measurements of round 2: PMC traffic -> profiles/traffic.json, the default bench line, per-algorithm lines, a traced run
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_26
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-kernel-events --no-miou"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$c -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 $Q > $OLDPWD/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/traffic.json && cp $OUT/traffic.json profiles/traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
# the driver's command line (defaults): roofline + cpu_baseline + mIoU legs
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; head -c 600 $OUT/bench_default.json; echo
timeout 300 python bench.py --steps 20 --warmup 3 $Q > $OUT/b_mt20.json 2>> $OUT/b.err
timeout 300 python bench.py --steps 20 --warmup 3 $Q > $OUT/b_mt20b.json 2>> $OUT/b.err
for a in suponly adv gct cct; do timeout 400 python bench.py --algo $a --steps 10 --warmup 3 $Q > $OUT/b_$a.json 2>> $OUT/b.err; done
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
tail -3 $OUT/b.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 $Q > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof -name "*results.db" | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py "$DB" $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-miou (round 2 final)" > /dev/null; python tools/prof_summary.py --one-step "$DB" $OUT/step_breakdown.txt > /dev/null; fi
rm -rf $OUT/prof
head -6 $OUT/step_breakdown.txt
