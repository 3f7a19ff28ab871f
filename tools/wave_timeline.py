#!/usr/bin/env python3
"""Where a wave of concurrent tasks spends its time: the LAST burst of device activity of a profiled executor_bench run (rocprofv3
--kernel-trace --memory-copy-trace --hip-runtime-trace), host and device side by side.
  * per HIP API function: calls, total and longest duration inside the burst, summed over all host threads;
  * per host thread: calls longer than <min_us>, in start order (the calls that hold a task thread);
  * the device: fraction of the burst with a kernel running / a copy running / neither, the idle gaps longer than 200 us, and per kernel name calls + total.
Usage: wave_timeline.py <dir with the csv files> [min_us=300]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
kern, cop, api = [], [], []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0].split("<")[0][-40:], r.get("Queue_Id", "?")))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Direction") or "")[:24]))
for f in glob.glob(root + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
kern_q = {(a, b, n): q for a, b, n, q in kern}
kern = [(a, b, n) for a, b, n, q in kern]
dev = sorted([(a, b) for a, b, _ in kern] + [(a, b) for a, b, _ in cop])
if not dev:
    sys.exit("no device activity")
# the last burst: device events less than 25 ms apart
lo = dev[-1][0]
hi = max(b for _, b in dev)
for a, b in reversed(dev[:-1]):
    if lo - b > 25_000_000:
        break
    lo = min(lo, a)
# the wave begins with its first host call: back up over host calls that end less than 3 ms before what we have
api.sort()
t0 = lo
for a, b, fn, th in reversed([x for x in api if x[0] < lo]):
    if t0 - b > 3_000_000:
        break
    t0 = min(t0, a)
t1 = max([hi] + [b for a, b, _, _ in api if a < hi + 3_000_000 and a >= t0])
print(f"# wave: {(t1 - t0) / 1e6:.2f} ms from its first host call to its last; device active from {(lo - t0) / 1e6:.2f} to {(hi - t0) / 1e6:.2f} ms")


def union(iv, a0, b0):
    tot, ce = 0, None
    cs = None
    for a, b in sorted(iv):
        a, b = max(a, a0), min(b, b0)
        if b <= a:
            continue
        if ce is None or a > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    if ce is not None:
        tot += ce - cs
    return tot


span = t1 - t0
kb = union([(a, b) for a, b, _ in kern], t0, t1)
cb = union([(a, b) for a, b, _ in cop], t0, t1)
ab = union([(a, b) for a, b, _ in kern] + [(a, b) for a, b, _ in cop], t0, t1)
print(f"# device: a kernel running {kb / span:.2f} of the wave, a copy running {cb / span:.2f}, either {ab / span:.2f}")
gaps, ce = [], t0
for a, b in sorted(x for x in dev if x[1] > t0 and x[0] < t1):
    if a - ce > 200_000:
        gaps.append((ce, a))
    ce = max(ce, b)
print("# device idle gaps > 0.2 ms: " + ("  ".join(f"{(a - t0) / 1e6:.2f}-{(b - t0) / 1e6:.2f}" for a, b in gaps) or "none"))
cbytes = defaultdict(lambda: [0, 0])
for a, b, d in cop:
    if a >= t0 and a < t1:
        cbytes[d][0] += 1
        cbytes[d][1] += b - a
for d, (n, ns) in cbytes.items():
    print(f"# copies {d}: {n} taking {ns / 1e6:.2f} ms in total")
ks = defaultdict(lambda: [0, 0])
for a, b, n in kern:
    if a >= t0 and a < t1:
        ks[n][0] += 1
        ks[n][1] += b - a
print("# kernels in the wave (calls, total ms):")
for n, (c, ns) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {n:42s} {c:5d} {ns / 1e6:8.3f}")
fs = defaultdict(lambda: [0, 0, 0])
for a, b, fn, th in api:
    if a >= t0 and a < t1:
        fs[fn][0] += 1
        fs[fn][1] += b - a
        fs[fn][2] = max(fs[fn][2], b - a)
print("# HIP API inside the wave, all threads (calls, total ms, longest ms):")
for fn, (c, ns, mx) in sorted(fs.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {fn:36s} {c:6d} {ns / 1e6:9.3f} {mx / 1e6:8.3f}")
per = defaultdict(list)
for a, b, fn, th in api:
    if a >= t0 and a < t1:
        per[th].append((a, b, fn))
print(f"# per host thread: calls of {min_us:g} us and more (start ms, duration ms, function); threads with HIP calls: {len(per)}")
for th, calls in sorted(per.items(), key=lambda kv: kv[1][0][0]):
    tot = sum(b - a for a, b, _ in calls)
    print(f"thread {th}: {len(calls)} calls, {tot / 1e6:.2f} ms inside HIP, first at {(calls[0][0] - t0) / 1e6:.2f}, last ends {(calls[-1][1] - t0) / 1e6:.2f}")
    for a, b, fn in calls:
        if (b - a) / 1e3 >= min_us:
            print(f"    {(a - t0) / 1e6:8.2f} {(b - a) / 1e6:8.2f}  {fn}")

# the device side in start order (kernels of 30 us and more, every copy): start ms, duration ms, what, hardware queue — written next to the summary when a third argument names a file
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        rows = [(a, b, n, kern_q.get((a, b, n), "?")) for a, b, n in kern if a >= t0 and a < t1 and b - a >= 30_000] + [(a, b, "copy " + d, "-") for a, b, d in cop if a >= t0 and a < t1]
        for a, b, n, q in sorted(rows):
            f.write(f"{(a - t0) / 1e6:8.3f} {(b - a) / 1e6:7.3f}  {n:40s} q{q}\n")
