#!/bin/bash
# one test N times with its assertion lines: tools/rep_one.sh "<pytest -k expression>" [N] [file]
cd "$(dirname "$0")/.."
K="${1:?-k expression}"; N="${2:-3}"; F="${3:-tests/test_multistep.py}"
for i in $(seq 1 $N); do python -m pytest $F -m gpu -q -s -p no:cacheprovider -k "$K" 2>&1 | grep "^E  \|passed\|failed\|worst\|cos\|ratio" | cut -c1-260 | tail -12; done
