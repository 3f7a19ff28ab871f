"""TPC-DS Q95 (BASELINE config 5) over two ranks on CPU (gloo): parallel.run_q95_distributed — web_sales and web_returns hash-exchanged on
the order number, stage A partition-local, the state rows merged by the Final aggregate on rank 0 — with the oracle standing in for the
per-rank GPU engine and the HIP partitioner (tests/exchange_helpers.py, test only).  The result must equal the direct set-based
evaluation of the query (tpcds.q95_reference), which shares no code with the plans."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from datafusion_comet_amd import native, parallel, tpcds
        from tests.exchange_helpers import OracleEngine, OraclePartitioner
        t = tpcds.q95_tables(3000, seed=95)
        # uneven, interleaved shards: rank r holds every row whose index ≡ r (mod world) of web_sales, a contiguous range of web_returns
        import pyarrow as pa
        import numpy as np
        ws = t["web_sales"].take(pa.array(np.arange(rank, t["web_sales"].num_rows, world)))
        wr = t["web_returns"].slice(*parallel.shard_range(t["web_returns"].num_rows, world, rank))
        mine = dict(t, web_sales=native.DeviceTable.from_arrow(ws, "cpu"), web_returns=native.DeviceTable.from_arrow(wr, "cpu"))
        timings = {}
        got = parallel.run_q95_distributed(OracleEngine(), OraclePartitioner(), mine, timings=timings)
        if rank == 0:
            want = tpcds.q95_reference(t)
            q.put(("ok", got == want, got[0], timings.get("exchange_rows", 0)))
        else:
            assert got is None
            q.put(("rank1", True, 0, 0))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_q95_equals_direct_evaluation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[0] != "err", r[1]
    ok = [r for r in res if r[0] == "ok"][0]
    assert ok[1] and ok[2] > 0 and ok[3] > 0
