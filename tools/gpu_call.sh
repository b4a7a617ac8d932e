#!/bin/bash
# ONE parametrised GPU-box session (replaces the per-call command files of rounds 1-2).  Everything lands in
# gpurun_out/<TAG>/; copy what should be judged into profiles/.
#
#   gpurun --timeout 900 -- 'tools/gpu_call.sh TAG STAGE [STAGE ...]'
#
# stages (run in order; a stage never aborts the following ones):
#   t=<pytest args>[@<-k expr>] python -m pytest <args> -m gpu -q [-k "<expr>"]   (e.g. t=tests/test_seam.py@mt+or+suponly)
#   all                        the whole GPU suite, -x
#   smoke                      __graft_entry__.smoke()
#   bench[=<bench.py args>]    python bench.py <args>                      -> bench_<n>.json   (default: driver defaults)
#   q[=<bench.py args>]        the timed line only (--no-cpu-baseline --no-kernel-events --no-miou --no-fp32-leg --no-fixture-parity)
#   ab=<ENV=V,ENV=V>[@<args>]  q-style run under environment switches      -> ab_<n>.json
#   prof[=<bench.py args>]     rocprofv3 --kernel-trace --stats of a q run + kernel_stats.csv + step_breakdown.txt
#   pmc=<counters>[@<args>]    one rocprofv3 --pmc pass (counters comma-separated; kernel-trace only) -> pmc_<n>/
#   traffic                    FETCH_SIZE / WRITE_SIZE passes of the MT step  -> traffic.json
#   py=<script and args>       python <script ...>                         -> py_<n>.log
#   sh=<command>               bash -c <command>                           -> sh_<n>.log
# "+" inside a stage argument stands for a space.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG="${1:?tag}"; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-kernel-events --no-miou --no-fp32-leg --no-fixture-parity"
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt
n=0
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    extra = ""
    if "roofline" in d: extra += " roofline.frac %s (%s)" % (d["roofline"]["frac"], d["roofline"]["kernel"])
    if "fp32_parity_mode" in d: extra += " fp32 %s img/s" % d["fp32_parity_mode"]["value"]
    print("%s: %s img/s %s ms/step step_mfma_frac %s%s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("step_mfma_frac"), extra))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for st in "$@"; do
  n=$((n+1))
  kind="${st%%=*}"; arg=""; [ "$st" != "$kind" ] && arg="${st#*=}"
  arg="${arg//+/ }"
  t0=$(date +%s)
  case $kind in
    t) files="${arg%%@*}"; kexpr=""; [ "$arg" != "$files" ] && kexpr="${arg#*@}"
       if [ -n "$kexpr" ]; then timeout 1500 python -m pytest $files -m gpu -q -s --tb=short -p no:cacheprovider -k "$kexpr" > $OUT/t_$n.log 2>&1; rc=$?
       else timeout 1500 python -m pytest $files -m gpu -q -s --tb=short -p no:cacheprovider > $OUT/t_$n.log 2>&1; rc=$?; fi
       grep -E "passed|failed|error|^FAILED|^ERROR" $OUT/t_$n.log | tail -8;;
    all) timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/test_all.log 2>&1; rc=$?
       grep -E "passed|failed|^FAILED|^ERROR" $OUT/test_all.log | tail -8;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -3 $OUT/smoke.log;;
    bench) timeout 1200 python bench.py $arg > $OUT/bench_$n.json 2> $OUT/bench_$n.err; rc=$?; line $OUT/bench_$n.json; tail -2 $OUT/bench_$n.err;;
    q) timeout 600 python bench.py $Q $arg > $OUT/q_$n.json 2> $OUT/q_$n.err; rc=$?; line $OUT/q_$n.json;;
    ab) envs="${arg%%@*}"; bargs=""; [ "$arg" != "$envs" ] && bargs="${arg#*@}"
       ( for kv in ${envs//,/ }; do export "$kv"; done; timeout 600 python bench.py $Q $bargs > $OUT/ab_$n.json 2> $OUT/ab_$n.err ); rc=$?
       echo -n "[$envs] "; line $OUT/ab_$n.json;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_$n -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 $Q $arg > $OLDPWD/$OUT/prof_$n.log 2>&1); rc=$?
       DB=$(find $OUT/prof_$n -name "*results.db" | head -1)
       if [ -n "$DB" ]; then
         python tools/prof_summary.py "$DB" $OUT/kernel_stats_$n.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 $Q $arg" | head -12
         python tools/prof_summary.py --one-step "$DB" $OUT/step_breakdown_$n.txt "one step of: python bench.py --steps 5 --warmup 2 $Q $arg" | head -24
       fi
       rm -rf $OUT/prof_$n;;
    pmc) ctr="${arg%%@*}"; bargs="--steps 1 --warmup 1"; [ "$arg" != "$ctr" ] && bargs="${arg#*@}"
       (cd /tmp && timeout 900 rocprofv3 --pmc ${ctr//,/ } --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$n -o p -- python $OLDPWD/bench.py $Q $bargs > $OLDPWD/$OUT/pmc_$n.log 2>&1); rc=$?
       python tools/pmc_summary.py $OUT/pmc_$n > $OUT/pmc_$n.txt 2>&1; head -30 $OUT/pmc_$n.txt;;
    traffic) for c in FETCH_SIZE WRITE_SIZE; do
         (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$c -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 $Q > $OLDPWD/$OUT/pmc_$c.log 2>&1); done
       python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/traffic.json "tools/gpu_call.sh $TAG traffic"; rc=$?
       rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE;;
    py) timeout 1200 python $arg > $OUT/py_$n.log 2>&1; rc=$?; tail -25 $OUT/py_$n.log;;
    sh) timeout 1200 bash -c "$arg" > $OUT/sh_$n.log 2>&1; rc=$?; tail -25 $OUT/sh_$n.log;;
    *) echo "unknown stage $st"; rc=64;;
  esac
  echo "== stage $n [$st] rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/stages.txt
done
