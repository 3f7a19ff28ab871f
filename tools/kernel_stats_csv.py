#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 `--kernel-trace --stats --output-format csv` file (<name>_kernel_stats.csv): short kernel name, calls,
total / average / min / max in µs, share.  Usage: tools/kernel_stats_csv.py <csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
print(f"{'kernel':36s} {'calls':>6s} {'total_us':>12s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for r in rows[:top]:
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", r["Name"])
    name = (m.group(1) if m else r["Name"])[:36]
    print(f"{name:36s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e3:12.1f} {float(r['AverageNs']) / 1e3:11.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
