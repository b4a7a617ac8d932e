#!/bin/bash
# round 4, call 2: kernel-argument prefetch + wait-free read-back passes; DMA stream variants (XCD order, K rotation, hot tile, 256-byte rows)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r04_02}; mkdir -p $OUT
export TMPDIR=/tmp
C=tools/cbench
timeout 300 $C --cfgs 17,18,24,25 --iters 20 --check > $OUT/base.txt 2>&1
for t in l3.1x1c:17 l3.1x1c:18 l3.3x3:18 l3.1x1b:25 l3.1x1b:18 l2.1x1b:17 l1.1x1b:25 l4.1x1b:25; do
  timeout 120 $C --trace $t >> $OUT/trace.txt 2>&1
done
timeout 300 $C --floor --only l3.1x1c,l3.1x1b,l2.1x1b,l4.1x1c,l1.1x1c > $OUT/floor.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 120 $C --trace l3.1x1c:18 > $OUT/trace_devkernarg1.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 120 $C --trace l3.1x1c:18 > $OUT/trace_devkernarg0.txt 2>&1
tail -3 $OUT/base.txt
