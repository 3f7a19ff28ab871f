#!/usr/bin/env python3
"""What HBM delivers to plain streaming kernels on this GPU: fill (write only), read-reduce (read only), copy (read + write) over 4 GiB, by
torch's own kernels and HIP events.  The rooflines the write-dominated kernels (Parquet decode to 16-byte decimals) should be priced against."""
import json
import sys

import torch

def timed(fn, reps=5):
    best = None
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best

n = (4 << 30) // 8
x = torch.empty(n, dtype=torch.int64, device="cuda:0")
y = torch.empty(n, dtype=torch.int64, device="cuda:0")
res = {}
ms = timed(lambda: x.fill_(7))
res["fill_TBps"] = n * 8 / ms / 1e9
ms = timed(lambda: x.zero_())
res["memset_TBps"] = n * 8 / ms / 1e9
ms = timed(lambda: y.copy_(x))
res["copy_TBps_read_plus_write"] = 2 * n * 8 / ms / 1e9
ms = timed(lambda: x.sum())
res["sum_TBps"] = n * 8 / ms / 1e9
xs = x.view(torch.int32)[: n]          # 2 GiB of int32 → 4 GiB... widen: read 4 B, write 16 B per element like a decimal decode
out = torch.empty((n // 2, 2), dtype=torch.int64, device="cuda:0")
def widen():
    out[:, 0] = xs[: n // 2]
ms = timed(widen)
res["note"] = "fill/memset: write only; sum: read only; copy: bytes read + bytes written"
print(json.dumps(res))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(json.dumps(res) + "\n")
