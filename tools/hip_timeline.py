#!/usr/bin/env python3
"""Host and device side by side for the LAST run of a profiled program: the HIP API calls (rocprofv3 --hip-runtime-trace) and the kernels / copies
(--kernel-trace, --memory-copy-trace when present) merged in start order from the last launch of <first_kernel> on, times relative to it.
Usage: hip_timeline.py <dir with the csv files> <first_kernel> [min_us=3] [nth_from_last=1]   — calls shorter than min_us are folded into a count; nth_from_last: start at
the n-th launch of <first_kernel> counted from the end (a run with five such launches: 5 = from its first).  A summary follows: device busy time, the gaps between
device events, the host calls by total time."""
import csv
import glob
import sys

root, first = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
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
s = starts[-min(nth, len(starts))]
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

# ---- summary of the window ----
from collections import Counter
win = ev[i0:]
gpu = sorted(e for e in win if e[2] == "GPU")
busy = sum(b - a for a, b, _, _ in gpu) / 1e3
span = (max(b for a, b, _, _ in win) - win[0][0]) / 1e3
gaps, end = [], gpu[0][1] if gpu else 0
for a, b, _, name in gpu[1:]:
    if a > end:
        gaps.append(((a - end) / 1e3, (end - t0) / 1e3, name))
    end = max(end, b)
print(f"\n== window {span:.1f} us, device busy {busy:.1f} us in {len(gpu)} events, {sum(g[0] for g in gaps):.1f} us of gaps between device events")
tot, cnt = Counter(), Counter()
for a, b, side, name in win:
    tot[(side, name)] += (b - a) / 1e3
    cnt[(side, name)] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {k[0]:4s} {k[1]:40s} x{cnt[k]:4d} {v:10.1f} us")
print("== largest gaps (us, at, before)")
for g in sorted(gaps, reverse=True)[:16]:
    print(f"  {g[0]:8.1f} at {g[1]:9.1f} before {g[2]}")
