#!/usr/bin/env python3
"""Device timeline of the LAST burst of activity in a rocprofv3 --kernel-trace --memory-copy-trace run: kernels and copies merged in start
order, times relative to the burst's first event, copies with their size and rate.  A burst = events separated by less than 5 ms.
Usage: timeline.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv
import re
import sys

ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"(sn2_\w+|zs2_\w+|pq_\w+|k_\w+|__amd\w+)", n)
    n = m.group(1)[:28] if m else n.split("(")[0].split("::")[-1][:28]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n, 0))
for r in csv.DictReader(open(sys.argv[2])):
    b = int(r.get("Bytes") or r.get("Size") or 0)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + (r.get("Direction") or r.get("Name") or "copy")[:28], b))
ev.sort()
# last burst
bursts, cur = [], []
for e in ev:
    if cur and e[0] - max(x[1] for x in cur) > 5_000_000:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
big = [b for b in bursts if any(x[2].startswith(("K sn2_", "K zs2_", "K pq_decode")) for x in b)] or [b for b in bursts if len(b) > 20]
b = big[-1] if big else bursts[-1]
# the last scan of the burst: from the last idle gap of > 1.5 ms before its first decompression / decode kernel
first = next((i for i in range(len(b) - 1, -1, -1) if b[i][2].startswith("K k_agg") and i < len(b) - 8), -1)
b = b[first + 1:]
t0 = b[0][0]
print(f"burst of {len(b)} events, {(max(x[1] for x in b) - t0) / 1e6:.2f} ms")
# aggregate consecutive events of the same name
agg = []
for s, e, n, sz in b:
    if agg and agg[-1][2] == n and s - agg[-1][1] < 300_000:
        agg[-1][1] = max(agg[-1][1], e)
        agg[-1][3] += sz
        agg[-1][4] += 1
        agg[-1][5] += e - s
    else:
        agg.append([s, e, n, sz, 1, e - s])
for s, e, n, sz, k, busy in agg:
    extra = f"  {sz / 1e6:8.1f} MB  {sz / max(busy, 1):6.1f} GB/s" if sz else ""
    print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  x{k:<4d} busy {busy / 1e6:7.3f} ms  {n}{extra}")
