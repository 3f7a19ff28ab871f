#!/usr/bin/env python3
"""Pre-flight for the in-library exchange (libcomet.so: partition kernels + RCCL ncclSend / ncclRecv groups): every rank creates the
communicator from the torch process group's rendezvous, exchanges a small table twice and checks that it received exactly the rows Spark's
HashPartitioning sends it.  bench.py runs this once (with a short timeout) before the multi-GPU legs and falls back to torch.distributed's
all_to_all for them if it fails or hangs — a leg must never stall the bench line.  One JSON line on rank 0."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import numpy as np
    import pyarrow as pa
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    import torch.distributed as dist
    from datafusion_comet_amd import native, parallel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    res = {"world": world, "ok": False}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    try:
        rng = np.random.default_rng(1000 + rank)
        t = pa.table({"k": pa.array(rng.integers(0, 1 << 40, a.rows), pa.int64()), "v": pa.array(rng.integers(0, 1 << 30, a.rows).astype(np.int32))})
        dt = native.DeviceTable.from_arrow(t, dev)
        if world > 1:
            ex = parallel.NativeExchange(parallel.native_comm_from_process_group(local))
            got = None
            for _ in range(2):
                got = ex.comm.exchange(dt, [0])
            # every received key hashes to this rank, and no row was lost or duplicated
            pids = native.partition_ids(got, [0], world)
            mine = bool((pids == rank).all().item()) if got.num_rows else True
            counts = torch.tensor([got.num_rows, a.rows], dtype=torch.int64, device=dev)
            dist.all_reduce(counts)
            res["ok"] = mine and int(counts[0].item()) == int(counts[1].item())
            flag = torch.tensor([1 if res["ok"] else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            res["ok"] = bool(flag.item())
        else:
            res["ok"] = True
    except Exception as e:      # noqa: BLE001
        res["error"] = repr(e)
    if rank == 0:
        line = json.dumps(res)
        print(line, flush=True)
        if a.out:
            with open(a.out, "w") as f:
                f.write(line + "\n")
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 3)


if __name__ == "__main__":
    main()
