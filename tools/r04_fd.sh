#!/bin/bash
# the GCT flaw detector's 4x4 convolutions alone (forward, data gradient, weight gradient), per tile configuration
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r04_fd}; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/pixelssl_amd:$LD_LIBRARY_PATH
timeout 600 tools/cbench --only fd --cfgs 1,8,9,10,17,18,20,21,24,25,28,29,30,31,34,35 --wcfgs 0,8,9,10,11,12,13 --modes fwd,dgrad,wgrad --iters 10 > $OUT/fd.txt 2>&1
