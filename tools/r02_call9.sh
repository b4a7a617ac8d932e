#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_9
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export PXL_STATS_REP=4 PXL_FUSE_BN_FINALIZE=1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events"
timeout 300 $B > $OUT/b_all.json 2> $OUT/b.err
PXL_SIDE_PRIO=0 timeout 300 $B > $OUT/b_noprio.json 2>> $OUT/b.err
PXL_PACK_STREAM=0 timeout 300 $B > $OUT/b_nopackstream.json 2>> $OUT/b.err
PXL_STATS_REP=2 timeout 300 $B > $OUT/b_rep2.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_losses"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
timeout 900 python -m pytest tests/test_multistep.py tests/test_gpu_net.py -m gpu -q -x --tb=short -p no:cacheprovider -k "mt_six or suponly_six or fixture or steps" > $OUT/ms.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/ms.log
tail -3 $OUT/b.err
