#!/bin/bash
# kernel-only durations of a conv_bench run (rocprofv3 --kernel-trace): usage tools/kt_bench.sh "<conv_bench args>"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$ROOT/gpurun_out/kt; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $ROOT/tools/conv_bench.py $1 > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, re
from collections import OrderedDict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    if "conv" not in n: continue
    k = (n, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc.setdefault(k, []).append(d)
for (n, g), v in acc.items():
    v = sorted(v)[: max(1, len(v) - 1)]          # drop the slowest (first, cold) launch
    print("%-62s grid %8s n=%3d  avg %7.2f us  min %7.2f" % (n, g, len(v), sum(v) / len(v), v[0]))
PY
