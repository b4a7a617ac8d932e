#!/bin/bash
# round 4, call 1: where does a contraction launch spend its time?  (tools/cbench: baseline, two streams, timelines, floors)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_01; mkdir -p $OUT
export TMPDIR=/tmp
C=tools/cbench
timeout 300 $C --cfgs -1,17,18,24,25 --iters 20 > $OUT/base.txt 2>&1
timeout 300 $C --cfgs 17,18,25 --iters 20 --dual --modes fwd --only l1.1x1b,l2.1x1b,l2.3x3,l3.1x1b,l3.1x1c,l3.3x3,l4.1x1b,l4.3x3d2 > $OUT/dual.txt 2>&1
for t in l3.1x1c:17 l3.1x1c:18:dual l3.3x3:18:dual l3.1x1b:25:dual l3.1x1b:18 l4.3x3d2:24 l2.1x1b:17 l1.1x1b:25 l4.1x1b:25; do
  timeout 120 $C --trace $t >> $OUT/trace.txt 2>&1
done
timeout 300 $C --floor --only l3.1x1c,l3.1x1b,l2.1x1b,l4.1x1c,l1.1x1c > $OUT/floor.txt 2>&1
tail -5 $OUT/base.txt; head -40 $OUT/trace.txt; head -30 $OUT/floor.txt
