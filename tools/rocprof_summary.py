#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result into a small text table: per-kernel calls/total/avg (µs).
Usage: tools/rocprof_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"{'kernel':48s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:48]:48s} {calls:7d} {total:12.3f} {avg:10.3f} {pct:7.2f}")
try:
    rows = list(db.execute("select name, value from counters_collection limit 0"))
except Exception:
    pass
