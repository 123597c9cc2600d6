#!/bin/bash
# Interleaved A/B of two builds of the library through their own bench.py (the driver's command, bare line): usage  ab_bench.sh <dirA> <dirB> [rounds] [extra bench args]
# A directory holds bench.py + pi-quant_amd/ (with the built libpiquant.so).  Prints per run: label, GiB/s, ms_per_step, kernel us (HIP events), roofline fraction.
A=$1; B=$2; R=${3:-5}; shift 3
for i in $(seq 1 $R); do
  for d in $A $B; do
    (cd $d && python bench.py --no-extras --steps 200 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])")
  done
done
