#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02_12
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_data.py -m gpu -q -x --tb=short -p no:cacheprovider -k "last_block or device_normalize" > $OUT/kern.log 2>&1; echo "kernels rc=$?"; tail -5 $OUT/kern.log
timeout 900 python -m pytest tests/test_multistep.py tests/test_gpu_net.py -m gpu -q -x --tb=short -p no:cacheprovider -k "mt_six or suponly_six or pspnet_sup or fixture or steps or bf16" > $OUT/ms.log 2>&1; echo "nets rc=$?"; tail -4 $OUT/ms.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-miou"
timeout 300 $B > $OUT/b_convfin.json 2> $OUT/b.err
PXL_CONV_FINALIZE=0 timeout 300 $B > $OUT/b_consumerfin.json 2>> $OUT/b.err
timeout 300 $B > $OUT/b_convfin2.json 2>> $OUT/b.err
for f in $OUT/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1], d["value"], d["ms_per_step"], d["final_losses"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 $OUT/b.err
