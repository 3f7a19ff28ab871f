"""The N>1 exchange path on CPU: two processes, gloo.  Each rank holds a row-range shard of customer / orders / lineitem;
parallel.run_q3_distributed runs the staged Q3 (three hash exchanges through torch.distributed.all_to_all_single) with the
oracle standing in for the per-rank GPU engine and for the HIP partitioner (tests/exchange_helpers.py, test only).
Checked: every row lands on the rank its Spark partition id names, nothing is lost or duplicated, validity bitmaps
survive the exchange, and the distributed top-10 equals the single-process oracle's."""
import os
import sys

import numpy as np
import pyarrow as pa
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
        from datafusion_comet_amd import native, parallel, serde as S, tpch
        from oracle import oracle as O
        from tests.exchange_helpers import OracleEngine, OraclePartitioner
        eng, part = OracleEngine(), OraclePartitioner()
        # ---- 1. exchange of a table with NULLs and a Boolean column
        rng = np.random.default_rng(100 + rank)
        n = 5000 + 37 * rank
        t = pa.table({"k": pa.array(rng.integers(0, 1000, n), pa.int64(), mask=rng.random(n) < 0.1),
                      "d": pa.array(rng.integers(-10**5, 10**5, n), pa.int32()).cast(pa.date32()),
                      "b": pa.array(rng.random(n) < 0.5, mask=(rng.random(n) < 0.2) if rank == 0 else None),
                      "m": pa.array([__import__("decimal").Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**9, 10**9, n)], pa.decimal128(12, 2)),
                      "s": pa.array(np.array(["", "a", "Customer#000000001", "x" * 70, "naïve ☕"], dtype=object)[rng.integers(0, 5, n)], pa.utf8(),
                                    mask=rng.random(n) < 0.15)})
        got = parallel.exchange(native.DeviceTable.from_arrow(t, "cpu"), [0], part).to_arrow()
        pids = O.hash_partition_ids(S, got, [0], world)
        ok_place = bool((pids == rank).all())
        rows = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows(t), rows(got)))
        before = sorted([r for g in gathered for r in g[0]], key=lambda r: tuple((x is None, str(x)) for x in r))
        after = sorted([r for g in gathered for r in g[1]], key=lambda r: tuple((x is None, str(x)) for x in r))
        ok_multiset = before == after
        # ---- 2. staged Q3 over shards == single-process Q3
        customer, orders, lineitem = tpch.q3_tables(6000, seed=3)
        sh = lambda tb: native.DeviceTable.from_arrow(tb.slice(*parallel.shard_range(tb.num_rows, world, rank)), "cpu")
        top, groups = parallel.run_q3_distributed(eng, part, sh(customer), sh(orders), sh(lineitem))
        total_groups = [None] * world
        dist.all_gather_object(total_groups, groups)
        if rank == 0:
            plan = tpch.q3_plan()
            partial = O.run_plan_to_arrow(S, plan, [customer, orders, lineitem])
            final = O.run_plan_to_arrow(S, S.final_of(plan, partial.schema), partial)
            want = parallel.q3_top10(final)
            q.put(("ok", ok_place, ok_multiset, top == want, sum(total_groups) == final.num_rows, len(want)))
        else:
            assert top is None
            if not (ok_place and ok_multiset):
                q.put(("err", "rank 1 placement/multiset check failed", 0, 0, 0, 0))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), 0, 0, 0, 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_and_staged_q3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == "ok", res[1]
    _, ok_place, ok_multiset, same_top, same_groups, nwant = res
    assert ok_place and ok_multiset
    assert same_top and same_groups and nwant == 10
