#!/bin/bash
# the row-streaming kernels ALONE, pure kernel durations (rocprofv3 kernel trace of tools/eltwise_bench.py: its own event timing is
# host-paced below ~9 us), per tensor shape of the trunk
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/elt -o e -- python $R/tools/eltwise_bench.py > /tmp/elt.log 2>&1)
cat /tmp/elt.log | grep -v amdgpu.ids
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/elt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# launches in order: per shape the bench runs red, app, apply_fwd, residual, relu_mask: 1 warm-up + 20 timed each
SHAPES = [(8*257*257,64),(8*129*129,64),(8*129*129,256),(8*65*65,128),(8*65*65,512),(8*33*33,256),(8*33*33,1024),(8*33*33,512),(8*33*33,2048)]
names = {'bn_bwd_reduce': 2, 'bn_bwd_apply_fused': 3, 'bn_apply_fwd': 2, 'residual_fwd': 3, 'relu_mask': 4}
seq = [r for r in rows if any(k in r['Kernel_Name'] for k in names)]
by = collections.OrderedDict()
idx = 0
print("%-14s %-22s %8s %8s %10s" % ("M x C", "kernel", "min us", "med us", "TB/s(med)"))
for (M, C) in SHAPES:
    for k, ntens in names.items():
        durs = []
        while idx < len(seq) and k in seq[idx]['Kernel_Name'] and len(durs) < 21:
            durs.append((int(seq[idx]['End_Timestamp']) - int(seq[idx]['Start_Timestamp'])) / 1e3); idx += 1
        if not durs: continue
        d = sorted(durs[1:] or durs)
        med = d[len(d)//2]
        print("%-14s %-22s %8.1f %8.1f %10.2f" % ("%dx%d" % (M, C), k, d[0], med, ntens * M * C * 2 / med / 1e6))
PY
