#!/bin/bash
# two consecutive PXL_DETERMINISTIC=1 runs of the multi-iteration parity cases (every workload): their printed figures must be identical
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r06_det}; mkdir -p $OUT
K="${2:-}"
for r in 1 2; do
  if [ -n "$K" ]; then PXL_DETERMINISTIC=1 timeout 1500 python -m pytest tests/test_multistep.py -m gpu -q -s -p no:cacheprovider -k "$K" > $OUT/run$r.log 2>&1
  else PXL_DETERMINISTIC=1 timeout 1500 python -m pytest tests/test_multistep.py -m gpu -q -s -p no:cacheprovider > $OUT/run$r.log 2>&1; fi
  grep -E "passed|failed" $OUT/run$r.log | tail -2
  grep -vE "^\.|passed|failed|warnings|^$|Warning|^  |seconds|^=|^-- Docs|parameters$" $OUT/run$r.log > $OUT/run${r}_figures.txt
done
if diff -q $OUT/run1_figures.txt $OUT/run2_figures.txt > /dev/null; then echo "IDENTICAL figures ($(wc -l < $OUT/run1_figures.txt) lines)"; else echo "figures DIFFER:"; diff $OUT/run1_figures.txt $OUT/run2_figures.txt | head -40; fi
