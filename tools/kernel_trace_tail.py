#!/usr/bin/env python3
"""Durations (ms) of the last launches of the kernels whose names contain one of the given substrings, in launch order, from a rocprofv3
--kernel-trace CSV.   Usage: kernel_trace_tail.py <kernel_trace.csv> <substring> [...]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in sys.argv[2:])]
for r in sel[-15:]:
    n = r["Kernel_Name"]
    n = n.split("::")[1][:24] if "::" in n else n[:24]
    print(f"{n:26s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.3f} ms  grid {r.get('Grid_Size_X')}  vgpr {r.get('VGPR_Count')}  lds {r.get('LDS_Block_Size')}")
