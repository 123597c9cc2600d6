"""Reduce rocprofv3 --pmc output (counter_collection CSVs) to per-launch HBM bytes of the quantize kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the fabric request counters; per
guides/MI355X_MICROARCH.md §HBM, on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced
stream (128-B requests tallied at 64 B) -> doubled here; WRITE_SIZE is uncalibrated and reported as is."""
import csv
import json
import sys
from pathlib import Path


def collect(root, counter):
    vals = []
    for f in Path(root).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and "pq::quantize_kernel<" in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {"launches_fetch": len(fetch), "launches_write": len(write)}
    if fetch:
        f = sorted(fetch)[len(fetch) // 2]
        out["FETCH_SIZE_median_raw"] = f
        out["fetch_bytes_per_launch_corrected"] = f * 1024 * 2
    if write:
        w = sorted(write)[len(write) // 2]
        out["WRITE_SIZE_median_raw"] = w
        out["write_bytes_per_launch"] = w * 1024
    if fetch and write:
        out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected"] + out["write_bytes_per_launch"]
        out["algorithmic_bytes_per_launch"] = 5 * 27264000
    print(json.dumps(out))


if __name__ == "__main__":
    main()
