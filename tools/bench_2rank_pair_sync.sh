#!/bin/bash
# bench.py as TWO ranks on ONE GPU: A/B of the paired statistics exchange (one peer exchange per BatchNorm for student + teacher,
# replicas folded in the exchange kernel: PXL_PAIR_SYNC=1, the default) against one fold + one exchange per network (=0), and
# against two passes on two streams (PXL_PAIR_FORWARD=0).  The ranks share the GPU: only the DIFFERENCES mean something.
cd "$(dirname "$0")/.."
ARGS="--steps 10 --warmup 3 --no-kernel-events --no-cpu-baseline --no-miou --no-fp32-leg --no-fixture-parity"
for sw in "PXL_PAIR_SYNC=1" "PXL_PAIR_SYNC=0" "PXL_PAIR_FORWARD=0"; do
  port=$((20000 + RANDOM % 20000))
  echo "== $sw"
  env $sw PXL_FORCE_DEVICE=0 PXL_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 2 $ARGS 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  %.2f ms/step  %.1f img/s  peer_contexts=%s paired_convs=%s paired_stat_exchanges=%s' % (d['ms_per_step'], d['value'], d.get('peer_contexts'), d.get('paired_convs'), d.get('paired_stat_exchanges')))"
done
