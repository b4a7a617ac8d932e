#!/bin/bash
# A/B of the contraction kernels against the round-3 library (tools/baseline/libpixelhip.so, built from 450ea78), same box, interleaved
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-r04_ab}; mkdir -p $OUT
ARGS="${2:---cfgs 17,18,24,25 --iters 20}"
for rep in 1 2; do
  LD_LIBRARY_PATH=$PWD/tools/baseline timeout 300 tools/cbench $ARGS > $OUT/base_$rep.txt 2>&1
  timeout 300 tools/cbench $ARGS > $OUT/new_$rep.txt 2>&1
done
python3 - $OUT <<'PY'
import sys,re,glob
out=sys.argv[1]
def parse(f):
    d={}
    for l in open(f):
        if '|' not in l or l.startswith('shape'): continue
        name=l.split()[0]
        for m in re.finditer(r'(fwd|dgrad|wgrad):(-?\d+)\s+([\d.]+) \(', l):
            d[(name,m.group(1),int(m.group(2)))]=float(m.group(3))
    return d
b=[parse(f) for f in sorted(glob.glob(out+'/base_*.txt'))]
n=[parse(f) for f in sorted(glob.glob(out+'/new_*.txt'))]
keys=sorted(set(b[0])&set(n[0]))
cnt={}
for l in open(sorted(glob.glob(out+'/new_*.txt'))[0]):
    pass
print("%-10s %-5s %3s %8s %8s %6s"%("shape","mode","cfg","base us","new us","ratio"))
tb={};tn={}
for k in keys:
    bb=min(x[k] for x in b if k in x); nn=min(x[k] for x in n if k in x)
    print("%-10s %-5s %3d %8.1f %8.1f %6.2f"%(k[0],k[1],k[2],bb,nn,nn/bb))
PY
