#!/bin/bash
# Round-2 GPU-box session.  Usage (repo root, via gpurun):  gpurun --timeout 2400 -- 'bash tools/gpu_session_r02.sh [phases...]'
#   phases: smoke newtests tests tune_fused tune_mm tune_half bench bench2 fit pmc prof rccl refbench hostcost soak matrix
# Everything lands under gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
PHASES="${*:-smoke newtests tests tune_fused tune_mm bench fit}"
echo "phases: $PHASES" | tee $OUT/session.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 | tee -a $OUT/session.log
nproc | tee -a $OUT/session.log; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket|NUMA" | tee -a $OUT/session.log
for ph in $PHASES; do
  echo "=== $ph $(date +%T)" | tee -a $OUT/session.log
  case $ph in
    smoke)    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/smoke.log ;;
    newtests) timeout 1200 python -m pytest tests/test_gpu_barrier.py tests/test_gpu_distributed.py -q -x -p no:cacheprovider > $OUT/pytest_new.log 2>&1; echo "newtests rc=$?" | tee -a $OUT/session.log; tail -30 $OUT/pytest_new.log ;;
    tests)    timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log; tail -30 $OUT/pytest_gpu.log ;;
    tune_fused) timeout 600 ./tools/tune_kernels 27264000 200 fused > $OUT/tune_fused.csv 2> $OUT/tune_fused.err; echo "tune_fused rc=$?" | tee -a $OUT/session.log; cat $OUT/tune_fused.csv; grep -E "differ|barrier wait|end \(" $OUT/tune_fused.err | head -40 ;;
    tune_mm)  timeout 600 ./tools/tune_kernels 27264000 200 mm2 > $OUT/tune_mm.csv 2> $OUT/tune_mm.err; echo "tune_mm rc=$?" | tee -a $OUT/session.log; sort -t, -k3 -n $OUT/tune_mm.csv | head -40 ;;
    tune_half) timeout 900 ./tools/tune_kernels 13632000 200 finals2 > $OUT/tune_half.csv 2> $OUT/tune_half.err; echo "tune_half rc=$?" | tee -a $OUT/session.log ;;
    bench)    timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/session.log; cat $OUT/bench.json; tail -5 $OUT/bench.err ;;
    bench2)   timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 200 --warmup 20 --backend gloo --share-gpu > $OUT/bench_n2_shared.json 2> $OUT/bench_n2_shared.err; echo "bench2 rc=$?" | tee -a $OUT/session.log; cat $OUT/bench_n2_shared.json ;;
    fit)      timeout 900 python tools/fit_fixed_cost.py > $OUT/fixed_cost_fit.json 2> $OUT/fixed_cost_fit.err; echo "fit rc=$?" | tee -a $OUT/session.log; python - <<'PY'
import json
d = json.load(open("gpurun_out/fixed_cost_fit.json"))
for k, v in d["kernels"].items():
    print(f"{k:34s} t0 {v['t0_us']:6.2f} us  BW {v['BW_GB/s']:7.1f} GB/s  frac@N1 {v['frac_at_27264000_measured']}  resid {v['max_residual_us']} us  n70 {v['numel_for_70_percent']}")
PY
      ;;
    pmc)      bash tools/pmc_all_kernels.sh 2>&1 | tail -160 ;;
    prof)     rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extras > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "prof rc=$?" | tee -a $OUT/session.log
              f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"; cat $OUT/prof_bench.json ;;
    rccl)     rm -rf $OUT/rccl_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$OUT/rccl_trace" -o rccl -- python "$OLDPWD/tools/rccl_single_rank_workload.py" > "$OLDPWD/$OUT/rccl_workload.log" 2> "$OLDPWD/$OUT/rccl_trace.err"); echo "rccl rc=$?" | tee -a $OUT/session.log
              f=$(find $OUT/rccl_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"; tail -5 $OUT/rccl_workload.log ;;
    hostcost) timeout 600 python tools/host_call_cost.py > $OUT/host_call_cost.json 2> $OUT/host_call_cost.err; echo "hostcost rc=$?" | tee -a $OUT/session.log; cat $OUT/host_call_cost.json ;;
    soak)     timeout $(( ${SOAK_SECONDS:-600} + 400 )) python tools/parity_soak.py --seconds ${SOAK_SECONDS:-600} --seed ${SOAK_SEED:-202} > $OUT/parity_soak_r02.json 2> $OUT/parity_soak_r02.err; echo "soak rc=$?" | tee -a $OUT/session.log; cat $OUT/parity_soak_r02.json ;;
    matrix)   timeout 900 python tools/dtype_matrix.py > $OUT/dtype_matrix.json 2> $OUT/dtype_matrix.err; echo "matrix rc=$?" | tee -a $OUT/session.log
              python -c "import json; d=json.load(open('$OUT/dtype_matrix.json')); [print(r['op'], r['in'], r['out'], r['mode'], r['us'], r['frac_of_peak']) for r in d['rows']]" ;;
    refbench) timeout 900 python tools/reference_style_benchmarks.py --plot $OUT/quant_benchmark.png > $OUT/reference_style.json 2> $OUT/reference_style.err; echo "refbench rc=$?" | tee -a $OUT/session.log ;;
  esac
done
echo "=== done $(date +%T)" | tee -a $OUT/session.log
