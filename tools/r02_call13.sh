#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_13
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "wgrad" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/kern.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
timeout 300 $B > $OUT/b_1.json 2> $OUT/b.err
timeout 300 $B > $OUT/b_2.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
# HBM-side traffic of the contraction kernels: two PMC passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && PXL_AUTOTUNE=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$c -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-miou > $OLDPWD/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
