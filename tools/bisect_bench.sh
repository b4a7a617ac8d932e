#!/bin/bash
# bench.py in every worktree under tmp_bisect/ (each built in place beforehand) and in the tree itself: which commit moved a number.
#   tools/bisect_bench.sh <bench args>        e.g. --algo adv --steps 20 --warmup 3
ARGS="${*:---algo adv --steps 20 --warmup 3}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for d in "$ROOT"/tmp_bisect/*/ "$ROOT"/; do
  [ -f "$d/bench.py" ] || continue
  for rep in 1 2; do
    (cd "$d" && python bench.py $ARGS --no-cpu-baseline --no-kernel-events --no-miou 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-28s %8.3f ms/step %8.1f img/s' % ('$(basename $d)', d['ms_per_step'], d['value']))")
  done
done
