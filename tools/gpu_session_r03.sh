#!/bin/bash
# Round-3 GPU-box session.  Usage (repo root, via gpurun):  gpurun --timeout 2400 -- 'bash tools/gpu_session_r03.sh [phases...]'
# Everything lands under gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
PHASES="${*:-smoke newtests tune_bf16 tune_f32var tune_mis bench}"
echo "phases: $PHASES" | tee $OUT/session.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 | tee -a $OUT/session.log
for ph in $PHASES; do
  echo "=== $ph $(date +%T)" | tee -a $OUT/session.log
  case $ph in
    smoke)    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/smoke.log ;;
    newtests) timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -q -x -p no:cacheprovider -m "gpu and not slow" -s > $OUT/pytest_new.log 2>&1; echo "newtests rc=$?" | tee -a $OUT/session.log; tail -30 $OUT/pytest_new.log ;;
    slowtests) timeout 1700 python -m pytest tests/test_gpu_distributed.py -q -x -p no:cacheprovider -m "slow" -s > $OUT/pytest_slow.log 2>&1; echo "slowtests rc=$?" | tee -a $OUT/session.log; tail -30 $OUT/pytest_slow.log ;;
    tests)    timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log; tail -30 $OUT/pytest_gpu.log ;;
    tune_bf16) TUNE_SETS=40 timeout 900 ./tools/tune_kernels 27264000 200 bf16 > $OUT/tune_bf16.csv 2> $OUT/tune_bf16.err; echo "tune_bf16 rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_bf16.csv ;;
    tune_f32var) TUNE_SETS=24 timeout 900 ./tools/tune_kernels 27264000 200 f32var > $OUT/tune_f32var.csv 2> $OUT/tune_f32var.err; echo "tune_f32var rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_f32var.csv ;;
    tune_mis) TUNE_SETS=24 timeout 900 ./tools/tune_kernels 27264000 200 mis > $OUT/tune_mis.csv 2> $OUT/tune_mis.err; echo "tune_mis rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_mis.csv ;;
    tune_dq)  TUNE_SETS=24 timeout 900 ./tools/tune_kernels 27264000 200 dq > $OUT/tune_dq.csv 2> $OUT/tune_dq.err; echo "tune_dq rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_dq.csv ;;
    tune_fused3) TUNE_SETS=24 timeout 900 ./tools/tune_kernels 27264000 200 fused3 > $OUT/tune_fused3.csv 2> $OUT/tune_fused3.err; echo "tune_fused3 rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_fused3.csv ;;
    tune_cap) TUNE_SETS=40 timeout 900 ./tools/tune_kernels 27264000 200 cap > $OUT/tune_cap.csv 2> $OUT/tune_cap.err; echo "tune_cap rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_cap.csv ;;
    tune_cap2) TUNE_SETS=40 timeout 900 ./tools/tune_kernels 27264000 200 cap2 > $OUT/tune_cap2.csv 2> $OUT/tune_cap2.err; echo "tune_cap2 rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_cap2.csv ;;
    tune_cap3) : > $OUT/tune_cap3.csv
               for n in 2097152 4194304 6815744 9000000 13632000 20000000 27264000 67108864 134217728; do
                 sets=$(( 3000000000 / (n * 5 / 2) )); [ $sets -gt 200 ] && sets=200; [ $sets -lt 8 ] && sets=8
                 echo "# numel $n sets $sets" >> $OUT/tune_cap3.csv
                 TUNE_SETS=$sets timeout 300 ./tools/tune_kernels $n 200 cap3 > $OUT/tune_cap3_one.csv 2>> $OUT/tune_cap3.err
                 echo "numel $n"; python tools/summarize_tune.py $OUT/tune_cap3_one.csv | cut -c1-150; cat $OUT/tune_cap3_one.csv >> $OUT/tune_cap3.csv
               done; echo "tune_cap3 rc=$?" | tee -a $OUT/session.log ;;
    tune_norm) TUNE_SETS=40 timeout 900 ./tools/tune_kernels 27264000 200 norm > $OUT/tune_norm.csv 2> $OUT/tune_norm.err; echo "tune_norm rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_norm.csv ;;
    tune_geo) TUNE_SETS=40 timeout 900 ./tools/tune_kernels 27264000 200 geo > $OUT/tune_geo.csv 2> $OUT/tune_geo.err; echo "tune_geo rc=$?" | tee -a $OUT/session.log
               python tools/summarize_tune.py $OUT/tune_geo.csv ;;
    xcd)      timeout 300 ./tools/diag_xcd_skew > $OUT/xcd_skew.txt 2>&1; timeout 300 ./tools/diag_xcd_skew 134217728 > $OUT/xcd_skew_2p27.txt 2>&1; echo "xcd rc=$?" | tee -a $OUT/session.log
              grep "launches\|first start ->" $OUT/xcd_skew.txt ;;
    bench)    timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/session.log; cut -c1-1500 $OUT/bench.json; tail -5 $OUT/bench.err ;;
    benchlong) timeout 900 python bench.py > $OUT/bench_long.json 2> $OUT/bench_long.err; echo "benchlong rc=$?" | tee -a $OUT/session.log; cut -c1-1500 $OUT/bench_long.json ;;
    wall)     timeout 600 python tools/diag_wall_overhead.py > $OUT/diag_wall_overhead.txt 2>&1; echo "wall rc=$?" | tee -a $OUT/session.log; cat $OUT/diag_wall_overhead.txt ;;
    matrix)   timeout 900 python tools/dtype_matrix.py > $OUT/dtype_matrix.json 2> $OUT/dtype_matrix.err; echo "matrix rc=$?" | tee -a $OUT/session.log
              python -c "import json; d=json.load(open('$OUT/dtype_matrix.json')); [print(r['op'], r['in'], r['out'], r['mode'], r['us'], r['frac_of_peak']) for r in d['rows']]" ;;
    fit)      timeout 900 python tools/fit_fixed_cost.py > $OUT/fixed_cost_fit.json 2> $OUT/fixed_cost_fit.err; echo "fit rc=$?" | tee -a $OUT/session.log
              python -c "
import json
d = json.load(open('gpurun_out/fixed_cost_fit.json'))
for k, v in d['kernels'].items():
    print(f\"{k:34s} t0 {v['t0_us']:6.2f} us  BW {v['BW_GB/s']:7.1f} GB/s  frac@N1 {v['frac_at_27264000_measured']}  resid {v['max_residual_us']} us\")
" ;;
    soak)     timeout $(( ${SOAK_SECONDS:-600} + 400 )) python tools/parity_soak.py --seconds ${SOAK_SECONDS:-600} --seed ${SOAK_SEED:-303} > $OUT/parity_soak_r03.json 2> $OUT/parity_soak_r03.err; echo "soak rc=$?" | tee -a $OUT/session.log; cat $OUT/parity_soak_r03.json ;;
    pmc)      bash tools/pmc_all_kernels.sh 2>&1 | tail -160 ;;
    prof)     rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extras > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "prof rc=$?" | tee -a $OUT/session.log
              f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"; cut -c1-1200 $OUT/prof_bench.json ;;
    *)        echo "unknown phase $ph" | tee -a $OUT/session.log ;;
  esac
done
echo "=== done $(date +%T)" | tee -a $OUT/session.log
