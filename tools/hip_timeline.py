#!/usr/bin/env python3
"""Host and device side by side for the LAST run of a profiled program: the HIP API calls (rocprofv3 --hip-runtime-trace) and the kernels / copies
(--kernel-trace, --memory-copy-trace when present) merged in start order from the last launch of <first_kernel> on, times relative to it.
Usage: hip_timeline.py <dir with the csv files> <first_kernel> [min_us=3]   — calls shorter than min_us are folded into a count"""
import csv
import glob
import sys

root, first = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU", r["Kernel_Name"].split("(")[0][-36:]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU", "copy " + (r.get("Direction") or "")[:24]))
for f in glob.glob(root + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host", r["Function"]))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2] == "GPU" and e[3] == first]
if not starts:
    sys.exit(f"no launch of {first}")
# the host call that launched it precedes it: back up to the previous host event more than 200 us before
s = starts[-1]
t0 = ev[s][0]
i0 = s
while i0 > 0 and t0 - ev[i0 - 1][0] < 200_000:
    i0 -= 1
small = 0
for a, b, side, name in ev[i0:]:
    d = (b - a) / 1e3
    if side == "host" and d < min_us:
        small += 1
        continue
    if small:
        print(f"{'':>10s}         ({small} host calls under {min_us:g} us)")
        small = 0
    print(f"{(a - t0) / 1e3:10.1f} us  {side:4s} {d:9.1f} us  {name}")
