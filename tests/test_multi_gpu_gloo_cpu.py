"""N>1 path on CPU: two processes, gloo backend — row-range sharding, gather of Partial states on rank 0, Final
merge.  There is no GPU here, so the oracle stands in for the per-rank engine (test only); what is under test is
the sharding/gather/merge logic of datafusion-comet_amd/parallel.py that bench.py --gpus N and a multi-GPU
deployment use."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_tile_the_table():
    from datafusion_comet_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1000, 59_986_052):
        for world in (1, 2, 3, 8):
            cover = 0
            for r in range(world):
                s, l = shard_range(n, world, r)
                assert s == cover and l >= 0
                cover += l
            assert cover == n
            sizes = [shard_range(n, world, r)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, which, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyarrow as pa
        from datafusion_comet_amd import parallel, serde as S, tpch
        from oracle import oracle as O
        from tests.test_final_agg_gpu import _final_plan
        if which == "q6":
            table, partial = tpch.lineitem_q6(40_000, seed=77), tpch.q6_plan()
        else:
            table, partial = tpch.lineitem_q1(30_000, seed=78), tpch.q1_plan()

        def run_partial(shard):
            return O.run_plan_to_arrow(S, partial, shard)

        def run_final(states):
            return O.run_plan_to_arrow(S, _final_plan(partial, states.schema), states)

        res = parallel.run_sharded_aggregate(table.num_rows, lambda s, l: table.slice(s, l), run_partial, run_final)
        if rank == 0:
            single = run_final(run_partial(table))
            key = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple(str(x) for x in r[:2]))
            q.put(("ok", key(res) == key(single), res.num_rows))
        else:
            assert res is None
    except Exception as e:  # pragma: no cover
        q.put(("err", repr(e), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["q6", "q1"])
def test_two_rank_sharded_aggregate_equals_single_process(which):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (0 if which == "q6" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, which, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, same, rows = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", same
    assert same
    assert rows == (1 if which == "q6" else 4)
