#!/bin/bash
# every bf16 GPU test three times in a row, all failures listed: flake hunt (the bf16 engine is not bitwise reproducible
# run to run -- fp32 atomics order the BatchNorm statistics -- so a bar that sits inside the spread fails now and then)
cd "$(dirname "$0")/.."
for i in 1 2 3; do python -m pytest tests -m gpu -q -p no:cacheprovider -k "bf16 or bfloat16" 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200; done
