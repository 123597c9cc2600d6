#!/usr/bin/env python3
"""Copies the judged summaries of a round's GPU sessions (tools/gpu_session.sh, tools/pmc_all_kernels.sh, the one-off tools) from
gpurun_out/ (scratch) into profiles/ (tracked), named <round>_*.  Run after the sessions:
    python tools/collect_profiles.py --round r04 [--since-minutes 720]
Only files written since the cutoff are taken (gpurun_out/ still holds what earlier rounds left there).

  bench.json                      -> <round>_bench.json (+ <round>_blocking_wait_ab.json: the three wait modes of a blocking call)
  prof/ (rocprofv3 --stats of bench.py)  -> <round>_bench_kernel_stats.csv, <round>_bench_kernel_stats_summary.json, <round>_bench_under_rocprof.json
  pmc_all/summary.json            -> <round>_pmc_all_kernels.json ; profiles/hbm_traffic.json (what bench.py reports as roofline.traffic)
  fixed_cost_fit.json             -> <round>_fixed_cost_fit.json
  host_call_cost.json             -> <round>_host_call_cost.json
  rccl_trace/*kernel_stats.csv    -> <round>_rccl_single_rank_kernel_stats.csv
  reference_style.json (+ png)    -> <round>_reference_style_benchmarks.json, <round>_quant_benchmark.png
  parity_soak.json                -> <round>_parity_soak_latest.json
  bench_n2_shared.json            -> <round>_bench_two_ranks_sharing_one_gpu.json
  tune_<mode>.csv                 -> <round>_tune_<mode>.csv (+ <round>_tune_<mode>_summary.txt from tools/summarize_tune.py)
  cpu_nt_stores.json, allreduce_cost.json, xcd_skew.txt, diag_wall_overhead.txt -> <round>_*
"""
import argparse
import csv
import json
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT, PROF = ROOT / "gpurun_out", ROOT / "profiles"
_ap = argparse.ArgumentParser()
_ap.add_argument("--round", default="r04")
_ap.add_argument("--since-minutes", type=float, default=720.0, help="take files of gpurun_out/ written in the last this many minutes")
_args = _ap.parse_args()
R = _args.round
CUTOFF = time.time() - 60.0 * _args.since_minutes


def fresh(p):
    return p.exists() and p.stat().st_size > 0 and p.stat().st_mtime >= CUTOFF


def copy_json(src, dst):
    p = OUT / src
    if fresh(p):
        json.loads(p.read_text())
        shutil.copy(p, PROF / dst)
        return [dst]
    return []


def stats_csv(src_glob, dst):
    stats = next(iter(sorted(p for p in OUT.glob(src_glob) if fresh(p))), None)
    if not stats:
        return None
    rows = list(csv.DictReader(stats.open()))
    with (PROF / dst).open("w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            name = r["Name"] if len(r["Name"]) < 240 else r["Name"][:200] + " ...<truncated>"
            w.writerow([name, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]), int(r["MinNs"]), int(r["MaxNs"]),
                        float(r["StdDev"])])
    return rows


def main():
    PROF.mkdir(exist_ok=True)
    done = []
    done += copy_json("bench.json", f"{R}_bench.json")
    done += copy_json("prof_bench.json", f"{R}_bench_under_rocprof.json")
    done += copy_json("fixed_cost_fit.json", f"{R}_fixed_cost_fit.json")
    done += copy_json("host_call_cost.json", f"{R}_host_call_cost.json")
    done += copy_json("reference_style.json", f"{R}_reference_style_benchmarks.json")
    done += copy_json("dtype_matrix.json", f"{R}_dtype_matrix.json")
    done += copy_json("parity_soak_r03.json", f"{R}_parity_soak_latest.json")   # the named runs (15 / 25 / 30 / 40 min) are copied by hand
    done += copy_json("bench_n2_shared.json", f"{R}_bench_two_ranks_sharing_one_gpu.json")
    if fresh(OUT / "quant_benchmark.png"):
        shutil.copy(OUT / "quant_benchmark.png", PROF / f"{R}_quant_benchmark.png")
        done.append(f"{R}_quant_benchmark.png")
    if fresh(OUT / "bench.json"):
        b = json.loads((OUT / "bench.json").read_text())
        bl = b.get("extras", {}).get("blocking_calls")
        if bl:
            (PROF / f"{R}_blocking_wait_ab.json").write_text(json.dumps({
                "what": "piquant_quantize fp32->uint8 at numel 27264000 on a blocking context (the reference's semantics: the call returns when its result is complete), "
                        "300 calls per mode on rotating buffers; kernel alone: see roofline.avg_launch_us",
                "kernel_avg_launch_us": b["roofline"]["avg_launch_us"], "modes": bl["by_wait_mode"], "default": bl["wait"],
                "modes_explained": {"sync": "hipStreamSynchronize", "write32": "hipStreamWriteValue32 of a sequence number into a pinned host-coherent word + host spin",
                                    "kernel": "the same word written by a one-thread kernel launched behind the work + host spin",
                                    "event": "the work kernel launched with a stop event (hipExtLaunchKernelGGL) + hipEventQuery spin: nothing enqueued behind the kernel"}}, indent=1) + "\n")
            done.append(f"{R}_blocking_wait_ab.json")
    rows = stats_csv("prof/**/*kernel_stats.csv", f"{R}_bench_kernel_stats.csv")
    if rows:
        digest = {"command": "rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras",
                  "kernels": [{"kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3), "min_us": round(int(r["MinNs"]) / 1e3, 3),
                               "max_us": round(int(r["MaxNs"]) / 1e3, 3), "percent": float(r["Percentage"])} for r in rows[:8]]}
        (PROF / f"{R}_bench_kernel_stats_summary.json").write_text(json.dumps(digest, indent=1) + "\n")
        done += [f"{R}_bench_kernel_stats.csv", f"{R}_bench_kernel_stats_summary.json"]
    if stats_csv("rccl_trace/**/*kernel_stats.csv", f"{R}_rccl_single_rank_kernel_stats.csv"):
        done.append(f"{R}_rccl_single_rank_kernel_stats.csv")
    pmc = OUT / "pmc_all" / "summary.json"
    if fresh(pmc):
        d = json.loads(pmc.read_text())
        (PROF / f"{R}_pmc_all_kernels.json").write_text(json.dumps({
            "command": "bash tools/pmc_all_kernels.sh: rocprofv3 --kernel-trace {--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (two groups)} -- python tools/config_kernels_workload.py "
                       "(200 stream-ordered launches of every kernel of the BASELINE configs at numel 27264000, rotating buffers); separate passes, kernel-trace only",
            "units": "per launch: median of the counters over the launches; fetch_MB = FETCH_SIZE KiB x 2 (gfx950: 128-B requests tallied at 64 B, guides/MI355X_MICROARCH.md HBM section), "
                     "write_MB = WRITE_SIZE KiB; avg/min/max_us from the --stats pass; X/SQ_WAVE_CYCLES = share of the waves' resident cycles",
            "kernels": d}, indent=1) + "\n")
        done.append(f"{R}_pmc_all_kernels.json")
        k = next((v for name, v in d.items() if name.startswith("pq::quantize_kernel<0, 8, 0,")), None)
        if k and "fetch_MB" in k and "write_MB" in k:
            (PROF / "hbm_traffic.json").write_text(json.dumps({"quantize_f32_u8": {
                "bytes_per_launch": round((k["fetch_MB"] + k["write_MB"]) * 1e6), "fetch_bytes": round(k["fetch_MB"] * 1e6), "write_bytes": round(k["write_MB"] * 1e6),
                "algorithmic_bytes": 136320000,
                "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel-trace only) over tools/config_kernels_workload.py; median over 200 launches at numel "
                          "27264000; FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B, guides/MI355X_MICROARCH.md HBM section), WRITE_SIZE KiB as reported; "
                          f"raw summaries in profiles/{R}_pmc_all_kernels.json"}}, indent=1) + "\n")
            done.append("hbm_traffic.json")
    for p in sorted(OUT.glob("tune_*.csv")):
        if fresh(p) and p.name != "tune_cap3_one.csv":
            shutil.copy(p, PROF / f"{R}_{p.name}")
            done.append(f"{R}_{p.name}")
            r = subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_tune.py"), str(p)], capture_output=True, text=True)
            if r.returncode == 0 and r.stdout.strip():
                (PROF / f"{R}_{p.stem}_summary.txt").write_text(r.stdout)
                done.append(f"{R}_{p.stem}_summary.txt")
    for src in ("cpu_nt_stores.json", "allreduce_cost.json", "xcd_skew.txt", "diag_wall_overhead.txt", "bench_long.json"):
        if fresh(OUT / src):
            shutil.copy(OUT / src, PROF / f"{R}_{src}")
            done.append(f"{R}_{src}")
    print("\n".join(done))


if __name__ == "__main__":
    main()
