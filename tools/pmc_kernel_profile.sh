#!/bin/bash
# Counter evidence for the headline kernel (separate --pmc passes, kernel-trace only): where do the wave cycles go?
#   gpurun -- 'bash tools/pmc_kernel_profile.sh'      -> gpurun_out/pmc_sq.json
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
pass() {  # name, counters...
  name=$1; shift
  rm -rf $OUT/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OLDPWD/$OUT/pmc_$name" -o pmc -- python "$OLDPWD/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2> "$OLDPWD/$OUT/pmc_$name.err")
  echo "pass $name rc=$?"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_SALU
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python - <<'PY'
import csv, json, collections
from pathlib import Path
res = {}
for d in ("sq1", "sq2", "tcc"):
    vals = collections.defaultdict(list)
    for f in Path(f"gpurun_out/pmc_{d}").rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "quantize_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        v.sort()
        res[k] = v[len(v) // 2]
res["launches_sampled"] = max((len(v) for v in vals.values()), default=0)
if "SQ_WAVE_CYCLES" in res and res["SQ_WAVE_CYCLES"]:
    wc = res["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in res:
            res[k + "_frac_of_wave_cycles"] = round(res[k] / wc, 4)
if res.get("TCC_HIT_sum") is not None and res.get("TCC_MISS_sum") is not None and (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]):
    res["L2_hit_rate"] = round(res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]), 4)
json.dump(res, open("gpurun_out/pmc_sq.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
