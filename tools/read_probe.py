#!/usr/bin/env python3
"""What bounds a Parquet scan before any byte reaches the GPU: the file's bytes come out of the page cache (pread → a copy by the kernel)
into pinned memory, then cross PCIe.  This probe times exactly that, with nothing else going on: T threads pread() disjoint 4 MiB pieces of
a file into (a) ordinary memory, (b) pinned memory (torch .pin_memory() = hipHostMalloc), and one hipMemcpyAsync per 64 MiB of the pinned
buffer to the device — the rates the scan's legs are priced against.  One JSON line."""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_all(fd, size, buf, threads, piece):
    nxt = [0]
    lock = threading.Lock()
    mv = memoryview(buf)

    def work():
        while True:
            with lock:
                off = nxt[0]
                nxt[0] += piece
            if off >= size:
                return
            n = min(piece, size - off)
            got = os.preadv(fd, [mv[off:off + n]], off)
            assert got == n

    ts = [threading.Thread(target=work) for _ in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", required=True)
    ap.add_argument("--threads", default="1,4,8,16,32,64")
    ap.add_argument("--piece", type=int, default=4 << 20)
    a = ap.parse_args()
    import numpy as np
    import torch
    size = os.path.getsize(a.file)
    fd = os.open(a.file, os.O_RDONLY)
    plain = np.empty(size, np.uint8)
    plain[:] = 0
    pinned_t = torch.empty(size, dtype=torch.uint8).pin_memory()
    pinned = pinned_t.numpy()
    pinned[:] = 0
    read_all(fd, size, plain, 16, a.piece)          # page cache warm
    res = {"file_bytes": size, "piece": a.piece, "cpus_visible": os.cpu_count(), "cpus_affinity": len(os.sched_getaffinity(0)), "pread_GBps": {}}
    try:
        res["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    for T in (int(x) for x in a.threads.split(",")):
        best_plain = min(read_all(fd, size, plain, T, a.piece) for _ in range(3))
        best_pin = min(read_all(fd, size, pinned, T, a.piece) for _ in range(3))
        res["pread_GBps"][str(T)] = {"to_ordinary_memory": round(size / best_plain / 1e9, 2), "to_pinned_memory": round(size / best_pin / 1e9, 2)}
    if torch.cuda.is_available():
        dev = torch.empty(size, dtype=torch.uint8, device="cuda")
        s = torch.cuda.Stream()
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s):
                for off in range(0, size, 64 << 20):
                    dev[off:off + (64 << 20)].copy_(pinned_t[off:off + (64 << 20)], non_blocking=True)
            s.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["h2d_GBps_64MiB_copies"] = round(size / best / 1e9, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
