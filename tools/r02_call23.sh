#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_23
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_s4l.py tests/test_gpu_kernels.py tests/test_parity_513.py -m gpu -q --tb=short -p no:cacheprovider -k "s4l or rotation or fused_task or pspnet-conditioned" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
run() { name=$1; algo=$2; shift; shift; env "$@" timeout 300 python bench.py --algo $algo $Q > $OUT/b_${algo}_$name.json 2>> $OUT/b.err; }
run default adv X=1
run nopack adv PXL_PACK_STREAM=0
run default gct X=1
run default cct X=1
run default suponly X=1
run default mt X=1
run nopack mt PXL_PACK_STREAM=0
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/ms.log
tail -3 $OUT/b.err
