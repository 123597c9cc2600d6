#!/usr/bin/env python3
"""Median / min over the interleaved passes of a tools/tune_kernels CSV, one line per variant."""
import csv
import statistics
import sys
from collections import OrderedDict

rows = OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.reader(f):
        if len(r) < 4 or r[0] in ("family",) or r[0].startswith("#"):
            continue
        if r[0] == "minmax_shifted":      # tune_kernels `mis`: the same scan with its input pointers moved by 4 bytes
            r = ["minmax+4B"] + r[2:]
        try:
            rows.setdefault((r[0], r[1]), []).append((float(r[2]), float(r[3])))
        except ValueError:
            pass
print(f"{'family':12s} {'variant':96s} {'med us':>8s} {'min us':>8s} {'GB/s(med)':>10s} {'frac':>6s}  n")
for (fam, var), v in rows.items():
    if fam == "check":     # correctness rows of the harness: the third column is 1 (equal) or 0, not a time
        print(f"{fam:12s} {var:96s} {'ok' if all(a == 1.0 for a, _ in v) else 'MISMATCH':>8s}")
        continue
    us = [a for a, _ in v if a > 0]
    if not us:
        continue
    med = statistics.median(us)
    gbs = v[0][1] * v[0][0] / med
    print(f"{fam:12s} {var:96s} {med:8.3f} {min(us):8.3f} {gbs:10.1f} {gbs / 8000:6.3f}  {len(us)}")
