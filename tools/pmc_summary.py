#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output dirs. Usage: pmc_summary.py <dir> [kernel-prefix]"""
import csv, glob, sys
root = sys.argv[1]
pref = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith(pref):
            continue
        a = acc.setdefault((k, r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
        acc.setdefault((k, "_VGPR"), [0.0, 0]); acc[(k, "_VGPR")][0] += float(r["VGPR_Count"]); acc[(k, "_VGPR")][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k[:40]:40s} {c:28s} {s / n:18.1f}  (n={n})")
