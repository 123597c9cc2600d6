#!/bin/bash
# One GPU-box session: smoke, GPU tests, kernel sweep, bench, rocprof.  Usage (from the repo root, via gpurun):
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh [phases...]'      phases: smoke tests tune bench prof profx pmc dyn
# Everything is written under gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
PHASES="${*:-smoke tests tune bench prof}"
echo "phases: $PHASES" | tee $OUT/session.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 | tee -a $OUT/session.log
nproc | tee -a $OUT/session.log; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" | tee -a $OUT/session.log

for ph in $PHASES; do
  echo "=== $ph $(date +%T)" | tee -a $OUT/session.log
  case $ph in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/smoke.log ;;
    tests) timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log; tail -40 $OUT/pytest_gpu.log ;;
    tune)  timeout 900 ./tools/tune_kernels 27264000 200 ${TUNE_ONLY:-all} > $OUT/tune.csv 2> $OUT/tune.err; echo "tune rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/tune.err ;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/session.log; cat $OUT/bench.json; tail -5 $OUT/bench.err ;;
    prof)  rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extras > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "prof rc=$?" | tee -a $OUT/session.log
           find $OUT/prof -name "*stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" ;;
    pmc)   rm -rf $OUT/pmc_fetch $OUT/pmc_write
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OLDPWD/$OUT/pmc_fetch" -o pmc -- python "$OLDPWD/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2> "$OLDPWD/$OUT/pmc_fetch.err"); echo "pmc fetch rc=$?" | tee -a $OUT/session.log
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OLDPWD/$OUT/pmc_write" -o pmc -- python "$OLDPWD/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2> "$OLDPWD/$OUT/pmc_write.err"); echo "pmc write rc=$?" | tee -a $OUT/session.log
           python tools/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err; cat $OUT/pmc_summary.json ;;
    profx) rm -rf $OUT/profx; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/profx" -o benchx -- python "$OLDPWD/bench.py" --steps 300 --warmup 30 --no-cpu-baseline > "$OLDPWD/$OUT/profx_bench.json" 2> "$OLDPWD/$OUT/profx.err"); echo "profx rc=$?" | tee -a $OUT/session.log ;;
    dyn)   rm -rf $OUT/dyn_stats $OUT/dyn_fetch $OUT/dyn_write
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/dyn_stats" -o dyn -- python "$OLDPWD/tools/dynamic_quantize_workload.py" > /dev/null 2> "$OLDPWD/$OUT/dyn_stats.err"); echo "dyn stats rc=$?" | tee -a $OUT/session.log
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OLDPWD/$OUT/dyn_fetch" -o dyn -- python "$OLDPWD/tools/dynamic_quantize_workload.py" > /dev/null 2> "$OLDPWD/$OUT/dyn_fetch.err"); echo "dyn fetch rc=$?" | tee -a $OUT/session.log
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OLDPWD/$OUT/dyn_write" -o dyn -- python "$OLDPWD/tools/dynamic_quantize_workload.py" > /dev/null 2> "$OLDPWD/$OUT/dyn_write.err"); echo "dyn write rc=$?" | tee -a $OUT/session.log
           python tools/summarize_pmc_by_kernel.py $OUT/dyn_fetch $OUT/dyn_write "$(find $OUT/dyn_stats -name '*kernel_stats.csv' | head -1)" > $OUT/dyn_summary.json 2> $OUT/dyn_summary.err; cat $OUT/dyn_summary.json ;;
  esac
done
echo "=== done $(date +%T)" | tee -a $OUT/session.log
