#!/bin/bash
# split-K of the sub-one-wave layer-3 grids through the EXISTING split-K path (atomics + finish kernel), no statistics:
# the upper bound of what a last-arriver variant could win before its epilogue is paid for
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for sk in 1 2 3 4; do echo "== --splitk $sk (forward, no statistics)"; timeout 120 tools/cbench --only l3.1x1c,l3.3x3,l3.1x1b,l4.1x1a,l4.1x1c --cfgs 16,17,18,24,25,34,35 --modes fwd --iters 20 --splitk $sk | grep "^l"; done
