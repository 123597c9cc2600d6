#!/bin/bash
# Interleaved A/B of builds / settings of the library through their own bench.py (the driver's command, bare line).
# usage: ab_bench.sh <rounds> <spec> [<spec> ...]      spec = dir[,ENV=VALUE...]   e.g.  ab_bench.sh 5 ab_old . .,PIQUANT_HIP_REFERENCE_LAYOUT=0
# A directory holds bench.py + pi-quant_amd/ (with the built libpiquant.so).  Prints per run: spec, GiB/s, ms_per_step, kernel us (HIP events), roofline fraction.
R=$1; shift
for i in $(seq 1 $R); do
  for spec in "$@"; do
    d=${spec%%,*}; envs=""; rest=${spec#"$d"}; rest=${rest#,}
    [ -n "$rest" ] && envs=$(echo "$rest" | tr ',' ' ')
    (cd $d && env $envs python bench.py --no-extras --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$spec', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])")
  done
done
