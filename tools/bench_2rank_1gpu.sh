#!/bin/bash
# bench.py as TWO ranks on ONE GPU (gloo rendezvous; RCCL refuses two ranks on one device): the A/B of the Sync-BN statistics
# exchange -- peer-mapped one-shot kernel (csrc/peer.hip) vs the torch.distributed callback -- on the real MT step.
# The two ranks share the GPU, so ms/step is NOT a throughput figure; the DIFFERENCE between the two lines is the exchange.
#   tools/bench_2rank_1gpu.sh [bench args]          (default: 4 + 4 images per rank at 513 x 513, 10 steps)
cd "$(dirname "$0")/.."
ARGS="${*:---steps 10 --warmup 3 --no-kernel-events --no-cpu-baseline --no-miou}"
for peer in 1 0; do
  port=$((20000 + RANDOM % 20000))
  echo "== PXL_PEER_SYNC=$peer"
  PXL_PEER_SYNC=$peer PXL_FORCE_DEVICE=0 PXL_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 2 $ARGS 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  %.2f ms/step  %.1f img/s  peer_contexts=%s rccl_ranks=%s grad_buckets=%s' % (d['ms_per_step'], d['value'], d.get('peer_contexts'), d.get('rccl_ranks'), d.get('grad_buckets')))"
done
