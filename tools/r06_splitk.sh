#!/bin/bash
# split-K of the sub-one-wave layer-3 / layer-4 grids, forward WITHOUT statistics (so without any last-arriver epilogue): the
# upper bound of what splitting K can win on these launches.  N = 1 is the same launch unsplit.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for sk in 1 2 3 4; do echo "== --splitk $sk: one partial-sum slab per slice (plain stores) + slab-summing finish kernel"; timeout 120 tools/cbench --only l3.1x1c,l3.3x3,l3.1x1b,l4.1x1a,l4.1x1c --cfgs 16,17,18,24,25,34,35 --modes fwd --iters 20 --splitk $sk | grep "^l" | grep -v s2; done
echo "== --splitk 2: ONE shared buffer, fp32 atomics (rounds 1-5)"; PXL_CBENCH_SPLITK_ATOMICS=1 timeout 120 tools/cbench --only l3.1x1c,l3.3x3 --cfgs 17,18,24 --modes fwd --iters 20 --splitk 2 | grep "^l" | grep -v s2
