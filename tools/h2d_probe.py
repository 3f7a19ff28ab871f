#!/usr/bin/env python3
"""What the PCIe link delivers to this process: pinned host → HBM copies of 16 / 64 / 256 MiB on 1, 2 and 4 streams, and the reverse
direction — the ceiling of the host ArrowArrayStream path (bench.py `paths.host_arrow_stream`) and of the Parquet scan's page upload.
One JSON line."""
import json
import time

import torch


def rate(total_mb, piece_mb, streams, d2h=False):
    dev = torch.device("cuda:0")
    n = total_mb // piece_mb
    host = [torch.empty(piece_mb << 20, dtype=torch.uint8).pin_memory() for _ in range(min(n, 8))]
    devb = [torch.empty(piece_mb << 20, dtype=torch.uint8, device=dev) for _ in range(min(n, 8))]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(ss[i % streams]):
                if d2h:
                    host[i % len(host)].copy_(devb[i % len(devb)], non_blocking=True)
                else:
                    devb[i % len(devb)].copy_(host[i % len(host)], non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, total_mb / 1024 * 1.073741824 / (time.perf_counter() - t0))
    return round(best, 1)


out = {}
for piece in (4, 16, 64, 256):
    for s in (1, 2, 4):
        out[f"h2d_{piece}MiB_x{s}"] = rate(2048, piece, s)
out["d2h_64MiB_x1"] = rate(2048, 64, 1, d2h=True)
out["d2h_64MiB_x2"] = rate(2048, 64, 2, d2h=True)
out["unit"] = "GB/s"
print(json.dumps(out))
