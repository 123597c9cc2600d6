#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE per kernel from rocprofv3 --pmc passes (one directory per counter), as HBM bytes per launch.
FETCH_SIZE KiB x2 on gfx950 (128-B requests tallied at 64 B, guides/MI355X_MICROARCH.md HBM section); WRITE_SIZE KiB as is.

  python tools/summarize_pmc_by_kernel.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [kernel-stats csv]
"""
import collections
import csv
import json
import sys
from pathlib import Path


def short(name):
    name = name.replace("void ", "")
    return name[: name.index("(")] if "(" in name else name


def medians(root, counter):
    vals = collections.defaultdict(list)
    for f in Path(root).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in vals.items()}


def all_counters(root):
    """--counters <dir>: every counter of every pass under <dir>/<pass>/ per pq:: kernel (median per launch), durations from the
    stats pass, HBM bytes from FETCH_SIZE (KiB x2 on gfx950) / WRITE_SIZE (KiB), wave-cycle fractions where SQ_WAVE_CYCLES exists."""
    root = Path(root)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in root.rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in root.rglob("*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    out = {}
    for k in sorted(set(vals) | set(dur)):
        if not k.startswith("pq::"):
            continue
        c = {n: sorted(v)[len(v) // 2] for n, v in vals.get(k, {}).items()}
        row = {}
        if k in dur:
            row.update({"launches": dur[k][1], "avg_us": round(dur[k][0], 2), "min_us": round(dur[k][2], 2), "max_us": round(dur[k][3], 2)})
        if "FETCH_SIZE" in c:
            row["fetch_MB"] = round(c["FETCH_SIZE"] * 2 * 1024 / 1e6, 2)
        if "WRITE_SIZE" in c:
            row["write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 2)
        if "fetch_MB" in row and "write_MB" in row and "avg_us" in row:
            row["hbm_GB/s_from_counters"] = round((row["fetch_MB"] + row["write_MB"]) / row["avg_us"] * 1e3, 1)
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if n in c:
                    row[n + "/SQ_WAVE_CYCLES"] = round(c[n] / wc, 4)
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
                if n in c:
                    row[n + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
        row["counters"] = {n: c[n] for n in sorted(c)}
        out[k] = row
    print(json.dumps(out, indent=1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--counters":
        return all_counters(sys.argv[2])
    fetch, write = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
    dur = {}
    if len(sys.argv) > 3:
        for r in csv.DictReader(open(sys.argv[3])):
            dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("pq::"):
            continue
        f = fetch.get(k, (0.0, 0))
        w = write.get(k, (0.0, 0))
        out[k] = {"launches": max(f[1], w[1]), "fetch_MB": round(f[0] * 2 * 1024 / 1e6, 2), "write_MB": round(w[0] * 1024 / 1e6, 2)}
        if k in dur:
            out[k]["avg_us"] = round(dur[k][0], 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
