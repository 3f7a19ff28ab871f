#!/usr/bin/env python3
"""rocprofv3's *_kernel_stats.csv as the short fixed-width table kept under profiles/: stats_table.py <kernel_stats.csv> [max rows] [skip-regex]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 24
skip = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
print(f"{'kernel':36s} {'calls':>6s} {'total_us':>12s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
n = 0
for r in rows:
    name = r["Name"]
    if skip and skip.search(name):
        continue
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", name)
    short = (m.group(1) if m else name)[:36]
    print(f"{short:36s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e3:12.1f} {float(r['AverageNs']) / 1e3:11.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
    n += 1
    if n >= limit:
        break
