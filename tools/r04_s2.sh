#!/bin/bash
# stride-2 data gradients: one launch per output-parity class (default) against the single launch that walks every tap
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r04_s2}; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/pixelssl_amd:$LD_LIBRARY_PATH
for sw in 1 0; do
  PXL_S2_CLASSES=$sw timeout 300 tools/cbench --only s2,.ds,fd.conv2,fd.conv3,fd.conv4 --cfgs 17,18,25,30 --modes dgrad --check --iters 10 > $OUT/bf16_classes$sw.txt 2>&1
  PXL_S2_CLASSES=$sw timeout 300 tools/cbench --f32 --only s2,.ds --cfgs 17,18,19 --modes dgrad --iters 5 > $OUT/f32_classes$sw.txt 2>&1
done
