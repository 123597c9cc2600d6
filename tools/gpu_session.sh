#!/bin/bash
# One GPU-box session (any round).  Usage (repo root, via gpurun):
#     gpurun --timeout 2400 -- 'ROUND=r04 bash tools/gpu_session.sh phase [phase ...]'
# Phases with an argument are written phase:arg[:arg], e.g.  tune:mm4:24  (tools/tune_kernels mode mm4, TUNE_SETS=24).
# Everything lands under gpurun_out/ (merged back by gpurun); tools/collect_profiles.py --round $ROUND copies the judged summaries to profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
ROUND="${ROUND:-r04}"
NUMEL="${NUMEL:-27264000}"
mkdir -p $OUT
PHASES="${*:-smoke tests bench}"
echo "round $ROUND phases: $PHASES" | tee $OUT/session.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 | tee -a $OUT/session.log
say() { echo "$*" | tee -a $OUT/session.log; }
for spec in $PHASES; do
  ph="${spec%%:*}"; rest="${spec#*:}"; [ "$rest" = "$spec" ] && rest=""
  a1="${rest%%:*}"; a2="${rest#*:}"; [ "$a2" = "$rest" ] && a2=""
  say "=== $spec $(date +%T)"
  case $ph in
    smoke)     timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; say "smoke rc=$?"; tail -3 $OUT/smoke.log ;;
    tests)     timeout 2700 python -m pytest tests -q -x -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1; say "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log ;;
    fasttests) timeout 2400 python -m pytest tests -q -x -m "gpu and not slow" -p no:cacheprovider --durations=10 ${a1:+-k "$a1"} > $OUT/pytest_fast.log 2>&1; say "fasttests rc=$?"; tail -40 $OUT/pytest_fast.log ;;
    tune)      # tune:<mode>[:sets]
               TUNE_SETS=${a2:-24} timeout 1200 ./tools/tune_kernels $NUMEL 200 $a1 > $OUT/tune_$a1.csv 2> $OUT/tune_$a1.err; say "tune $a1 rc=$?"
               python tools/summarize_tune.py $OUT/tune_$a1.csv | cut -c1-230 ;;
    xcd)       timeout 300 ./tools/diag_xcd_skew > $OUT/xcd_skew.txt 2>&1; say "xcd rc=$?"; grep "launches\|first start ->" $OUT/xcd_skew.txt ;;
    bench)     timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; say "bench rc=$?"; cut -c1-1800 $OUT/bench.json; tail -5 $OUT/bench.err ;;
    benchlong) timeout 900 python bench.py > $OUT/bench_long.json 2> $OUT/bench_long.err; say "benchlong rc=$?"; cut -c1-1500 $OUT/bench_long.json ;;
    bench2)    # the N = 2 control flow on one GPU: bench.py starts its own two ranks (gloo, sharing the device)
               timeout 1200 python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 > $OUT/bench_n2_shared.json 2> $OUT/bench_n2_shared.err; say "bench2 rc=$?"
               cut -c1-3000 $OUT/bench_n2_shared.json; tail -5 $OUT/bench_n2_shared.err ;;
    wall)      timeout 600 python tools/diag_wall_overhead.py > $OUT/diag_wall_overhead.txt 2>&1; say "wall rc=$?"; cat $OUT/diag_wall_overhead.txt ;;
    matrix)    timeout 900 python tools/dtype_matrix.py > $OUT/dtype_matrix.json 2> $OUT/dtype_matrix.err; say "matrix rc=$?"
               python -c "import json; d=json.load(open('$OUT/dtype_matrix.json')); [print(r['op'], r['in'], r['out'], r['mode'], r['us'], r['frac_of_peak']) for r in d['rows']]" ;;
    fit)       timeout 900 python tools/fit_fixed_cost.py > $OUT/fixed_cost_fit.json 2> $OUT/fixed_cost_fit.err; say "fit rc=$?" ;;
    soak)      timeout $(( ${SOAK_SECONDS:-600} + 400 )) python tools/parity_soak.py --seconds ${SOAK_SECONDS:-600} --seed ${SOAK_SEED:-404} > $OUT/parity_soak.json 2> $OUT/parity_soak.err; say "soak rc=$?"; cut -c1-1500 $OUT/parity_soak.json ;;
    refstyle)  timeout 900 python tools/reference_style_benchmarks.py --plot $OUT/quant_benchmark.png > $OUT/reference_style.json 2> $OUT/reference_style.err; say "refstyle rc=$?"; cut -c1-1500 $OUT/reference_style.json; tail -3 $OUT/reference_style.err ;;
    cpunt)     timeout 900 python tools/diag_cpu_nt_stores.py > $OUT/cpu_nt_stores.json 2> $OUT/cpu_nt_stores.err; say "cpunt rc=$?"; cat $OUT/cpu_nt_stores.json ;;
    arcost)    timeout 900 python tools/all_reduce_compute_cost.py > $OUT/allreduce_cost.json 2> $OUT/allreduce_cost.err; say "arcost rc=$?"; cut -c1-2000 $OUT/allreduce_cost.json ;;
    pmc)       bash tools/pmc_all_kernels.sh 2>&1 | tail -160 ;;
    prof)      rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extras > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); say "prof rc=$?"
               f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"; cut -c1-1200 $OUT/prof_bench.json ;;
    py)        # py:<script under tools/>: any one-off tool, output to gpurun_out/<script>.out
               timeout 1200 python tools/$a1 > $OUT/${a1%.py}.out 2> $OUT/${a1%.py}.err; say "py $a1 rc=$?"; cut -c1-3000 $OUT/${a1%.py}.out | tail -40 ;;
    *)         say "unknown phase $spec" ;;
  esac
done
say "=== done $(date +%T)"
