#!/bin/bash
# build: ab_quant_kernel.sh build <label> <header dir> [extra hipcc flags]   -> ab_bin/<label>      (here, cross-compiled)
# run:   ab_quant_kernel.sh run <rounds> <label> [<label> ...]               (on the GPU box: alternates the executables)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  label=$2; dir=$3; shift 3
  mkdir -p ab_bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -std=c++17 -O3 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=12 -Iinclude -I$dir "$@" ${AB_SRC:-tools/ab_quant_kernel.hip} -o ab_bin/$label 2>&1 | grep -v "warning\|^$" || true
  ls -la ab_bin/$label
else
  rounds=$2; shift 2
  for i in $(seq 1 $rounds); do for l in "$@"; do ./ab_bin/$l 27264000 2000 $l $AB_THREADS | tail -${AB_TAIL:-3}; done; done
fi
