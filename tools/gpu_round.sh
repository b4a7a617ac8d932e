#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [stage ...]   stages: kernels net rest all smoke bench benchfast algos prof eltwise convbench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGES="${*:-kernels net smoke bench}"
echo "stages: $STAGES" | tee $OUT/stages.txt
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/gpu.txt
for s in $STAGES; do
  case $s in
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/test_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $OUT/stages.txt; tail -30 $OUT/test_kernels.log;;
    kernels_all) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line -p no:cacheprovider > $OUT/test_kernels.log 2>&1; echo "kernels_all rc=$?" | tee -a $OUT/stages.txt; tail -60 $OUT/test_kernels.log;;
    net) timeout 1200 python -m pytest tests/test_gpu_net.py -m gpu -q -s --tb=short -p no:cacheprovider > $OUT/test_net.log 2>&1; echo "net rc=$?" | tee -a $OUT/stages.txt; grep -vE "^\s*$" $OUT/test_net.log | tail -60;;
    rest) timeout 1500 python -m pytest tests/test_psp.py tests/test_cct.py tests/test_adv.py tests/test_cutmix.py tests/test_gct.py tests/test_gct_flawmap.py tests/test_gpu_dist.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_rest.log 2>&1; echo "rest rc=$?" | tee -a $OUT/stages.txt; grep -E "passed|failed|^FAILED" $OUT/test_rest.log | tail -5;;
    all) timeout 1800 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/test_all.log 2>&1; echo "all rc=$?" | tee -a $OUT/stages.txt; grep -E "passed|failed|^FAILED" $OUT/test_all.log | tail -5;;
    eltwise) timeout 300 python tools/eltwise_bench.py > $OUT/eltwise_bench.txt 2>&1; echo "eltwise rc=$?" | tee -a $OUT/stages.txt; cat $OUT/eltwise_bench.txt;;
    algos) for al in suponly adv gct cct; do timeout 600 python bench.py --algo $al --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$al.log 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/bench_$al.log').readline()); print('$al', d['value'], d['ms_per_step'], d.get('step_mfma_frac'))"; done;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/stages.txt; tail -5 $OUT/smoke.log;;
    bench) timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/stages.txt; tail -3 $OUT/bench.log; tail -5 $OUT/bench.err;;
    benchfast) timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "benchfast rc=$?" | tee -a $OUT/stages.txt; tail -3 $OUT/bench.log; tail -5 $OUT/bench.err;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o mt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?" | tee -a $OUT/stages.txt; db=$(find $OUT/prof -name "*results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py "$db" $OUT/prof/kernel_stats.csv | head -25 && python tools/prof_summary.py --one-step "$db" $OUT/prof/step_breakdown.txt | head -8;;
    convbench) timeout 600 python tools/conv_bench.py --dtype bf16 --cfgs=-1,0,1,2,4,5,6 > $OUT/convbench_bf16.log 2>&1; echo "convbench rc=$?" | tee -a $OUT/stages.txt; cat $OUT/convbench_bf16.log;;
    convbench32) timeout 600 python tools/conv_bench.py --dtype fp32 --cfgs=-1 > $OUT/convbench_fp32.log 2>&1; echo "convbench32 rc=$?" | tee -a $OUT/stages.txt; cat $OUT/convbench_fp32.log;;
  esac
done
