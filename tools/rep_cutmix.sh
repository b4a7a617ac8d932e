#!/bin/bash
# the bf16 CutMix six-iteration test six times in a row: the run-to-run spread of its smallest loss term
cd "$(dirname "$0")/.."
for i in 1 2 3 4 5 6; do python -m pytest tests/test_multistep.py -m gpu -q -s -p no:cacheprovider -k "cutmix_six and bf16" 2>&1 | grep "iter 5\|passed\|failed"; done
