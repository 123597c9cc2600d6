#!/bin/bash
# Counter evidence for every kernel of the BASELINE configs (separate --pmc passes, kernel-trace only -- never combined with
# sys/hip/hsa tracing): per-kernel duration, HBM bytes fetched / written, and where the wave cycles go.
#   gpurun -- 'bash tools/pmc_all_kernels.sh'      -> gpurun_out/pmc_all/summary.json
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_all
rm -rf $OUT; mkdir -p $OUT
W="$PWD/tools/config_kernels_workload.py"
pass() {  # name, rocprof args...
  name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace "$@" -f csv -d "$OLDPWD/$OUT/$name" -o p -- python "$W" > /dev/null 2> "$OLDPWD/$OUT/$name.err")
  echo "pass $name rc=$?"
}
pass stats --stats
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
python tools/summarize_pmc_by_kernel.py --counters $OUT > $OUT/summary.json 2> $OUT/summary.err
cat $OUT/summary.json | head -150
