#!/usr/bin/env python3
"""Copies the judged summaries of one tools/gpu_session.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

  python tools/collect_profiles.py r01        # after: gpurun -- 'bash tools/gpu_session.sh smoke tests bench prof pmc'

bench.json -> profiles/<round>_bench.json, the rocprofv3 --stats table of the same command -> <round>_bench_kernel_stats.csv
(+ a short JSON digest with kernel names cut to a readable length), the PMC summary -> <round>_pmc_summary.json and
profiles/hbm_traffic.json (what bench.py reports as roofline.traffic).
"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT, PROF = ROOT / "gpurun_out", ROOT / "profiles"


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    PROF.mkdir(exist_ok=True)
    done = []
    for src, dst in (("bench.json", f"{rnd}_bench.json"), ("prof_bench.json", f"{rnd}_bench_under_rocprof.json")):
        if (OUT / src).exists() and (OUT / src).stat().st_size:
            json.loads((OUT / src).read_text())          # must be one valid JSON document
            shutil.copy(OUT / src, PROF / dst)
            done.append(dst)
    stats = next(iter(sorted(OUT.glob("prof/**/*kernel_stats.csv"))), None)
    if stats:
        rows = list(csv.DictReader(stats.open()))
        with (PROF / f"{rnd}_bench_kernel_stats.csv").open("w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                name = r["Name"] if len(r["Name"]) < 240 else r["Name"][:200] + " ...<truncated>"
                w.writerow([name, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]), int(r["MinNs"]),
                            int(r["MaxNs"]), float(r["StdDev"])])
        digest = {"command": "rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras",
                  "kernels": [{"kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                               "min_us": int(r["MinNs"]) / 1e3, "max_us": int(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:8]]}
        (PROF / f"{rnd}_bench_kernel_stats_summary.json").write_text(json.dumps(digest, indent=1))
        done.append(f"{rnd}_bench_kernel_stats.csv")
    pmc = OUT / "pmc_summary.json"
    if pmc.exists() and pmc.stat().st_size:
        s = json.loads(pmc.read_text())
        shutil.copy(pmc, PROF / f"{rnd}_pmc_summary.json")
        traffic = {"quantize_f32_u8": {
            "bytes_per_launch": s["hbm_bytes_per_launch"], "fetch_bytes": s["fetch_bytes_per_launch_corrected"], "write_bytes": s["write_bytes_per_launch"],
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel-trace only) over bench.py; median over the timed "
                      "launches; FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B, guides/MI355X_MICROARCH.md HBM section), WRITE_SIZE KiB as "
                      f"reported; raw summaries in profiles/{rnd}_pmc_summary.json"}}
        (PROF / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1))
        done.append("hbm_traffic.json")
    statsx = next(iter(sorted(OUT.glob("profx/**/*kernel_stats.csv"))), None)
    if statsx:
        rows = [r for r in csv.DictReader(statsx.open()) if r["Name"].startswith(("void pq::", "pq::"))]
        with (PROF / f"{rnd}_all_kernels_stats.csv").open("w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for r in rows:
                w.writerow([r["Name"][:160], int(r["Calls"]), float(r["AverageNs"]), int(r["MinNs"]), int(r["MaxNs"]), float(r["Percentage"])])
        done.append(f"{rnd}_all_kernels_stats.csv")
    dyn = OUT / "dyn_summary.json"
    if dyn.exists() and dyn.stat().st_size:
        doc = {"command": "rocprofv3 --kernel-trace {--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} -- python tools/dynamic_quantize_workload.py "
                          "(60 fused calls, then 60 calls with fusion off; fp32 -> uint8, numel 27 264 000, 6 rotating buffer sets)",
               "units": "per launch: median HBM MB from the counters (FETCH_SIZE KiB x2 on gfx950, WRITE_SIZE KiB), average duration from --stats",
               "kernels": json.loads(dyn.read_text())}
        (PROF / f"{rnd}_dynamic_quantize_pmc.json").write_text(json.dumps(doc, indent=1))
        done.append(f"{rnd}_dynamic_quantize_pmc.json")
    print("updated:", ", ".join(done))


if __name__ == "__main__":
    main()
