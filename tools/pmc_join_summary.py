#!/usr/bin/env python3
"""Per-dispatch PMC view of the hash-join kernels of one Q95 run: every rocprofv3 pass directory under <root> (q95_fetch, q95_tcc,
q95_sq: --pmc … --kernel-trace) lists its k_jbuild / k_jprobe / k_jlds dispatches in launch order — the same program, so ordinal k of
one pass is ordinal k of the others — with its counters; q95_stats gives the durations.  FETCH_SIZE is KiB and is doubled (gfx950
counts 64 B per 128-B request, MI355X_MICROARCH.md); WRITE_SIZE is KiB.   Usage: pmc_join_summary.py <root> [prefix=q95]"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
PFX = sys.argv[2] if len(sys.argv) > 2 else "q95"
KERNELS = ("k_jbuild", "k_jprobe", "k_jprobe_km", "k_jprobe_b", "k_jprobe_bkm", "k_jprobe_bm", "k_jdprobe", "k_jlds", "k_jphist", "k_jpscat", "k_jtbuild", "k_filter", "k_jbmap", "k_jbcnt",
           "k_jdrows", "k_gagg", "k_gemit", "k_jbemit", "k_jbcount", "k_pack", "k_emit")


def short(n):
    return n.split("(")[0]


durs = defaultdict(list)
for f in glob.glob(root + f"/{PFX}_stats/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if k in KERNELS:
            durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
per = defaultdict(lambda: defaultdict(dict))      # kernel → ordinal → counter → value
for d in sorted(glob.glob(root + f"/{PFX}_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        disp = OrderedDict()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k not in KERNELS:
                continue
            key = int(r["Dispatch_Id"])
            disp.setdefault(key, (k, {}))[1][r["Counter_Name"]] = disp.get(key, (k, {}))[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp[key][1]["_grid"] = r.get("Grid_Size", "")
        ordinal = defaultdict(int)
        for key in sorted(disp):
            k, c = disp[key]
            per[k][ordinal[k]].update(c)
            ordinal[k] += 1
for k in KERNELS:
    if k not in per:
        continue
    print(f"== {k}: {len(per[k])} dispatches in the profiled program (warm-up + timed runs)")
    cols = sorted({c for o in per[k].values() for c in o if not c.startswith("_")})
    print("  ord " + " ".join(f"{c[:14]:>14s}" for c in ["us", "grid"] + cols) + "   HBM_rd_MB  hit%")
    for o in sorted(per[k]):
        c = per[k][o]
        us = durs[k][o] if o < len(durs[k]) else float("nan")
        vals = [f"{us:14.1f}", f"{c.get('_grid', ''):>14s}"] + [f"{c.get(x, float('nan')):14.0f}" for x in cols]
        rd = 2 * 1024 * c.get("FETCH_SIZE", 0) / 1e6
        h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        print(f"  {o:3d} " + " ".join(vals) + f"   {rd:9.1f}  {100 * h / max(h + m, 1):5.1f}")
