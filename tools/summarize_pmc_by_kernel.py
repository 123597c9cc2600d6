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


def main():
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
