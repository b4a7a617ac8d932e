#!/bin/bash
# fp32 LDS-DMA kernels alone (torch-free): per-shape time / TFLOP/s of every tile configuration, and two streams at once
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r04_f32}; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/pixelssl_amd:$LD_LIBRARY_PATH
timeout 600 tools/cbench --f32 --cfgs 1,8,9,10,11,16,17,18,19 --modes fwd,dgrad --iters 10 > $OUT/f32_fwd_dgrad.txt 2>&1
timeout 600 tools/cbench --f32 --wcfgs 0,8,9,10,11,12,13 --modes wgrad --iters 10 > $OUT/f32_wgrad.txt 2>&1
timeout 600 tools/cbench --f32 --cfgs 17,18,19 --modes fwd --dual --iters 10 > $OUT/f32_dual.txt 2>&1
tail -2 $OUT/f32_fwd_dgrad.txt $OUT/f32_wgrad.txt
