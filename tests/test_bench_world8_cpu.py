"""Multi-GPU first contact without a GPU: everything of `bench.py --gpus 8` and its legs that can fail on an 8-GPU box OTHER than the wire itself.

* bench.py's rank plumbing as plain functions, driven with recording stand-ins for torch / torch.distributed: every one of eight ranks binds LOCAL_RANK's device BEFORE
  a communicator is created; the child legs of one rank set get ONE rendezvous port per leg, distinct across legs and from the parent's; the fallback → exit-code rule.
* EIGHT gloo processes on the CPU run what the eight ranks of the bench run around their kernels (datafusion-comet_amd/parallel.py): the headline's row-range sharding +
  gather of Partial states + Final on rank 0 (Q1), the hash exchange (a table with THREE distinct keys: most (rank → partition) pairs are empty — nothing is posted for
  them and nothing is lost), the staged Q3 with its three exchanges and the Q95 plan over exchanged web_sales / web_returns — the oracle standing in for the per-rank
  engine and the HIP partitioner (tests/exchange_helpers.py, test only).  Results must equal the single-process evaluation.
The wire itself at eight ranks — libcomet's RCCL transport against a strict stand-in librccl.so: call sequence, grouped send / recv, ncclCommCount = 8, no send or recv posted
for an empty partition — is tests/test_rccl_shim_procs_cpu.py (world 8 is one of its parameters)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_world8", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Recorder:
    def __init__(self):
        self.calls = []

    class _Cuda:
        def __init__(self, outer):
            self.outer = outer

        def set_device(self, d):
            self.outer.calls.append(("set_device", d))

    @property
    def cuda(self):
        return _Recorder._Cuda(self)

    def device(self, s):
        return s

    def init_process_group(self, backend, **kw):
        self.calls.append(("init_process_group", backend, kw.get("device_id")))


def test_every_rank_binds_its_device_before_the_communicator_exists():
    b = _bench()
    for local in range(8):
        rec = _Recorder()
        dev = b.init_rank(rec, rec, local, 8)
        assert dev == f"cuda:{local}"
        assert rec.calls == [("set_device", local), ("init_process_group", "nccl", f"cuda:{local}")], rec.calls
    rec = _Recorder()
    assert b.init_rank(rec, rec, 0, 1) == "cuda:0" and rec.calls == [("set_device", 0)]      # one GPU: no process group at all


def test_child_legs_meet_on_their_own_ports():
    b = _bench()
    parent = {"MASTER_PORT": "29511", "MASTER_ADDR": "10.0.0.7", "TORCHELASTIC_RUN_ID": "x", "GROUP_RANK": "0", "ROLE_RANK": "3", "PATH": "/bin"}
    offsets = [917, 1017, 1517, 2017, 2117, 3017, 4017, 5017, 5117, 5217, 5267, 5317, 5417, 5517]      # the legs' offsets in bench.py
    src = open(os.path.join(ROOT, "bench.py")).read()
    import re
    used = [int(x) for x in re.findall(r"world, (?:args\.leg_timeout|\d+), (\d+)(?: \+ \([^)]*\))?\)", src)]
    assert sorted(set(used)) == sorted(set(used) | set()) and len(used) >= 10 and set(used) <= set(offsets) | {5017}, used
    ports = {}
    for off in offsets:
        envs = [b.leg_env(parent, r, r, 8, off) for r in range(8)]
        assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["MASTER_ADDR"] == "127.0.0.1"
        assert [e["RANK"] for e in envs] == [str(r) for r in range(8)] and all(e["WORLD_SIZE"] == "8" and e["LOCAL_RANK"] == e["RANK"] for e in envs)
        assert not any(k in envs[0] for k in ("TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")) and envs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports[off] = envs[0]["MASTER_PORT"]
    assert len(set(ports.values())) == len(offsets) and "29511" not in ports.values()


def test_fallback_becomes_the_exit_code():
    b = _bench()
    ok = {"q3": {"exchange_transport": "rccl"}, "q95": {"exchange_transport": "rccl"}}
    assert b.multi_gpu_exit_code(1, True, True, "torch-fallback", ok, False) == 0            # one GPU: nothing to fall back from
    assert b.multi_gpu_exit_code(8, True, True, "native", ok, False) == 0
    assert b.multi_gpu_exit_code(8, True, True, "torch-fallback", ok, False) == 4            # the probe failed
    assert b.multi_gpu_exit_code(8, True, True, "torch-fallback", ok, True) == 0             # --allow-fallback
    assert b.multi_gpu_exit_code(8, True, True, "native", dict(ok, q95={"exit_code": 4}), False) == 4      # a leg fell back on its own
    assert b.multi_gpu_exit_code(8, True, True, "native", None, False) == 0                  # ranks other than 0 know the probe's verdict only
    assert b.multi_gpu_exit_code(8, False, True, "torch-fallback", None, False) == 0         # --no-extra-legs: no exchange ran
    assert b.multi_gpu_exit_code(8, True, False, "torch-fallback", ok, False) == 0           # both exchange legs switched off
    line = b.compact_line({"metric": "m", "value": 1.0, "unit": "rows/s", "n_gpus": 8, "config": {}, "roofline": {},
                           "q3": {"sec_per_run": 0.004, "n_gpus": 8, "exchange_transport": "rccl", "roofline": {}}})
    assert line["n_gpus"] == 8 and line["legs"]["q3_n_gpus"] == 8 and line["legs"]["q3_exchange_transport"] == "rccl"


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    try:
        import pyarrow as pa
        b = _bench()
        rec = _Recorder()

        class _Dist:      # the real process group, created where bench.py creates it; the device binding recorded in front of it
            @staticmethod
            def init_process_group(backend, **kw):
                rec.calls.append(("init_process_group", backend))
                dist.init_process_group(backend, rank=rank, world_size=world)

        assert b.init_rank(rec, _Dist, rank, world, backend="gloo") == f"cuda:{rank}"
        assert rec.calls == [("set_device", rank), ("init_process_group", "gloo")]
        from datafusion_comet_amd import native, parallel, serde as S, tpch, tpcds
        from oracle import oracle as O
        from tests.exchange_helpers import OracleEngine, OraclePartitioner
        from tests.test_final_agg_gpu import _final_plan
        eng, part = OracleEngine(), OraclePartitioner()
        out = {}
        # 1. the headline's shape: contiguous row ranges, Partial per rank, states gathered on rank 0, Final there
        table, partial = tpch.lineitem_q1(12_000, seed=81), tpch.q1_plan()
        res = parallel.run_sharded_aggregate(table.num_rows, lambda s, l: table.slice(s, l), lambda sh: O.run_plan_to_arrow(S, partial, sh),
                                             lambda st: O.run_plan_to_arrow(S, _final_plan(partial, st.schema), st))
        if rank == 0:
            single = O.run_plan_to_arrow(S, _final_plan(partial, O.run_plan_to_arrow(S, partial, table).schema), O.run_plan_to_arrow(S, partial, table))
            key = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple(str(x) for x in r[:2]))
            out["q1"] = key(res) == key(single) and res.num_rows == 4
        else:
            assert res is None
        # 2. an exchange in which most (rank → partition) pairs are EMPTY: three distinct keys over eight partitions, and rank 5 holds no rows at all
        rng = np.random.default_rng(300 + rank)
        n = 0 if rank == 5 else 400 + rank
        t = pa.table({"k": pa.array(rng.integers(0, 3, n) * 1_000_003, pa.int64()), "v": pa.array(rng.integers(-1000, 1000, n), pa.int32(), mask=rng.random(n) < 0.1)})
        got = parallel.exchange(native.DeviceTable.from_arrow(t, "cpu"), [0], part).to_arrow()
        pids = O.hash_partition_ids(S, got, [0], world)
        rows = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows(t), rows(got), got.num_rows))
        before = sorted((r for g in gathered for r in g[0]), key=lambda r: tuple((x is None, str(x)) for x in r))
        after = sorted((r for g in gathered for r in g[1]), key=lambda r: tuple((x is None, str(x)) for x in r))
        out["exchange"] = bool((pids == rank).all()) and before == after and sum(1 for g in gathered if g[2] == 0) >= 5      # ≤ 3 ranks receive anything
        # 3. the staged Q3 (three exchanges) over eight shards
        customer, orders, lineitem = tpch.q3_tables(2500, seed=3)
        sh = lambda tb: native.DeviceTable.from_arrow(tb.slice(*parallel.shard_range(tb.num_rows, world, rank)), "cpu")
        top, groups = parallel.run_q3_distributed(eng, part, sh(customer), sh(orders), sh(lineitem))
        all_groups = [None] * world
        dist.all_gather_object(all_groups, groups)
        if rank == 0:
            plan = tpch.q3_plan()
            p1 = O.run_plan_to_arrow(S, plan, [customer, orders, lineitem])
            final = O.run_plan_to_arrow(S, S.final_of(plan, p1.schema), p1)
            out["q3"] = top == parallel.q3_top10(final) and sum(all_groups) == final.num_rows and len(top) == 10
        else:
            assert top is None
        # 4. Q95 over exchanged fact tables, dimensions on every rank
        tt = tpcds.q95_tables(3000, seed=95)
        ws = tt["web_sales"].take(pa.array(np.arange(rank, tt["web_sales"].num_rows, world)))
        wr = tt["web_returns"].slice(*parallel.shard_range(tt["web_returns"].num_rows, world, rank))
        got95 = parallel.run_q95_distributed(eng, part, dict(tt, web_sales=native.DeviceTable.from_arrow(ws, "cpu"), web_returns=native.DeviceTable.from_arrow(wr, "cpu")))
        if rank == 0:
            out["q95"] = got95 == tpcds.q95_reference(tt)
            out["q95_count"] = got95[0]
        else:
            assert got95 is None
        q.put(("ok", rank, out))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_ranks_run_the_bench_legs_host_logic_end_to_end():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[0] == "ok", r[2]
    assert sorted(r[1] for r in res) == list(range(world))
    r0 = [r for r in res if r[1] == 0][0][2]
    assert r0.pop("q95_count") > 0
    assert r0 == {"q1": True, "exchange": True, "q3": True, "q95": True}, r0
    for r in res:
        if r[1] != 0:
            assert r[2] == {"exchange": True}, r
