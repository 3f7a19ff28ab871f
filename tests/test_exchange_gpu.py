"""GPU parity for the exchange step and device-resident stage outputs:
comet_partition_indices / comet_take_column against the oracle's restatement of the reference's shuffle-writer scratch
computation (bit-exact, same order), hash partition ids against the oracle, comet_execute_plan_device against the
host path, and the staged Q3 (three exchanges) against the single-plan Q3 and the oracle — on one GPU with the ranks
simulated in-process (slices moved with torch.cat) and with a 1-rank RCCL group."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, parallel, serde as S, tpch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,P", [(0, 4), (1, 1), (63, 3), (8192, 8), (100_003, 8), (250_000, 200), (70_001, 1024)])
def test_partition_indices_match_reference_restatement(built, n, P):
    import torch
    from oracle import oracle as O
    rng = np.random.default_rng(n + P)
    pids = rng.integers(0, P, n).astype(np.int32)
    if n > 1000:
        pids[: n // 3] = 0          # a skewed stretch
    d = torch.from_numpy(pids).cuda()
    starts = torch.empty(P + 1, dtype=torch.int64, device="cuda")
    idx = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    rc = native.lib().comet_partition_indices(d.data_ptr() if n else None, n, P, starts.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, native.lib().comet_last_error(0)
    want_starts, want_idx = O.partition_starts_and_indices(pids, P)
    assert starts.cpu().tolist() == want_starts.tolist()
    assert np.array_equal(idx.cpu().numpy().view(np.uint32)[:n], want_idx)


def test_partition_indices_reference_example_and_bad_id(built):
    import torch
    import json, os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["partition_indices"]
    d = torch.tensor(k["partition_ids"], dtype=torch.int32, device="cuda")
    starts = torch.empty(k["num_partitions"] + 1, dtype=torch.int64, device="cuda")
    idx = torch.empty(len(k["partition_ids"]), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert native.lib().comet_partition_indices(d.data_ptr(), d.numel(), k["num_partitions"], starts.data_ptr(), idx.data_ptr(), st) == 0
    assert idx.cpu().tolist() == k["partition_row_indices"] and starts.cpu().tolist() == k["partition_starts"]
    assert native.lib().comet_partition_indices(d.data_ptr(), d.numel(), 3, starts.data_ptr(), idx.data_ptr(), st) == -2
    assert b"outside" in native.lib().comet_last_error(0)


@pytest.mark.parametrize("width", [0, 1, 2, 4, 8, 16])
def test_take_column(built, width):
    import torch
    rng = np.random.default_rng(width)
    n_src, n = 50_000, 33_333
    idx = rng.integers(0, n_src, n).astype(np.uint32)
    if width == 0:
        bits = rng.integers(0, 2, n_src).astype(np.uint8)
        src = np.packbits(bits, bitorder="little")
        want = np.packbits(bits[idx], bitorder="little")
    else:
        src = rng.integers(0, 256, n_src * width).astype(np.uint8)
        want = src.reshape(n_src, width)[idx].reshape(-1)
    dsrc, didx = torch.from_numpy(src).cuda(), torch.from_numpy(idx.view(np.int32)).cuda()
    out = torch.zeros(len(want), dtype=torch.uint8, device="cuda")
    assert native.lib().comet_take_column(width, dsrc.data_ptr(), didx.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def _rows(t):
    if t is None:
        return []
    return sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))


def test_hash_partitioner_places_rows_like_spark(built):
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    n = 200_000
    from decimal import Decimal
    t = pa.table({"k": pa.array(rng.integers(-10**12, 10**12, n), pa.int64(), mask=rng.random(n) < 0.05),
                  "d": pa.array(rng.integers(0, 20000, n), pa.int32()).cast(pa.date32()),
                  "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
                  "m": tpch._dec128_array(rng.integers(0, 10**9, n), 12, 2),
                  "s": pa.array(np.array(["", "a", "Customer#000000001", "x" * 70, "naïve ☕"], dtype=object)[rng.integers(0, 5, n)], pa.utf8(),
                                mask=rng.random(n) < 0.15)})
    dt = native.DeviceTable.from_arrow(t)
    for keys in ([0], [1, 0], [3]):
        pids = native.partition_ids(dt, keys, 8)
        want = O.hash_partition_ids(S, t, keys, 8)
        assert np.array_equal(pids.cpu().numpy(), want)
    part, starts = parallel.HipPartitioner()(dt, [0], 8)
    want_pids = O.hash_partition_ids(S, t, [0], 8)
    ws, wi = O.partition_starts_and_indices(want_pids, 8)
    assert starts == ws.tolist()
    assert part.to_arrow().equals(t.take(pa.array(wi)))


def test_execute_plan_device_equals_host_path(built):
    t = tpch.lineitem_q6(300_000, seed=4)
    D = tpch.DEC
    plan = S.project(S.filter_(S.scan([D, D, D, S.T_DATE]), S.lt(S.col(0, D), S.lit(2400, D))),
                     [S.col(3, S.T_DATE), S.math("add", S.col(1, D), S.col(2, D), S.decimal(13, 2)), S.col(0, D)])
    host = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 3, plan.encode(), batch_size=0))
    for inp in (native.HostInput.from_table(t), native.DeviceInput(native.DeviceTable.from_arrow(t))):
        dev = native.execute_to_device([inp], 3, plan.encode())
        assert dev.num_rows == host.num_rows > 1000
        got = dev.to_arrow()
        assert [f.type for f in got.schema] == [f.type for f in host.schema]
        assert got.rename_columns(host.column_names).equals(host)
    # a stream whose schema is not what the Scan declares is rejected (the reference would cast, scan.rs:134-164)
    bad = S.project(S.scan([S.T_DATE, D, D, D]), [S.col(0, S.T_DATE)])
    with pytest.raises(native.CometNativeException, match="declares"):
        native.execute_to_device([native.HostInput.from_table(t)], 1, bad.encode())
    # an ungrouped aggregate (one row) is exported through the host call only
    with pytest.raises(native.CometNativeException, match="ungrouped"):
        native.execute_to_device([native.HostInput.from_table(t)], 2, tpch.q6_plan().encode())


def test_grouped_aggregate_states_stay_on_device(built):
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n = 400_000
    t = pa.table({"k": pa.array(rng.integers(0, 150_000, n), pa.int64()), "d": pa.array(rng.integers(9000, 9100, n), pa.int32()).cast(pa.date32()),
                  "m": tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2)})
    D = tpch.DEC
    plan = S.hash_agg(S.scan([S.T_INT64, S.T_DATE, D]), [S.col(0, S.T_INT64), S.col(1, S.T_DATE)],
                      [S.sum_(S.col(2, D), S.decimal(22, 2)), S.count(S.col(2, D))])
    dev = native.execute_to_device([native.DeviceInput(native.DeviceTable.from_arrow(t))], 5, plan.encode())
    want = O.run_plan_to_arrow(S, plan, t)
    assert dev.num_rows == want.num_rows > 300_000          # high cardinality: the global table grows several times
    assert _rows(dev.to_arrow()) == _rows(want)
    # and the resident states feed the Final stage directly
    fplan = S.final_of(plan, dev.schema)
    got = pa.Table.from_batches(native.execute_to_table([native.DeviceInput(dev)], 4, fplan.encode(), batch_size=0))
    assert _rows(got) == _rows(O.run_plan_to_arrow(S, fplan, want))
    # no input rows → an empty resident table
    empty = native.execute_to_device([native.HostInput.from_table(t.slice(0, 0))], 5, plan.encode())
    assert empty.num_rows == 0


def test_device_output_of_join_feeds_next_plan(built):
    customer, orders, lineitem = tpch.q3_tables(20_000, seed=5)
    st = tpch.q3_stage_plans()
    eng = parallel.GpuEngine()
    c = eng.run_device(st["customer"][0], [customer], 1)
    o = eng.run_device(st["orders"][0], [orders], 4)
    j1 = eng.run_device(st["join1"][0], [c, o], 3)
    l = eng.run_device(st["lineitem"][0], [lineitem], 3)
    partial = eng.run_host(st["join2agg"][0], [j1, l], tpch.Q3_NUM_OUTPUT_COLS)
    from oracle import oracle as O
    want = O.run_plan_to_arrow(S, tpch.q3_plan(), [customer, orders, lineitem])
    assert _rows(partial) == _rows(want)


def _simulated_ranks_q3(world, customer, orders, lineitem):
    """All ranks of a `world`-way run executed one after the other on this GPU; the all-to-all becomes torch.cat of slices."""
    import torch
    st = tpch.q3_stage_plans()
    eng, part = parallel.GpuEngine(), parallel.HipPartitioner()

    def shard(tb, r):
        return tb.slice(*parallel.shard_range(tb.num_rows, world, r))

    def exchange_all(outs, key):
        parts = [part(o, [key], world) for o in outs]
        res = []
        for dst in range(world):
            vals = []
            for ci in range(len(outs[0].values)):
                w = native.value_width(outs[0].schema.field(ci).type)
                vals.append(torch.cat([p.values[ci].reshape(-1, w)[s[dst]:s[dst + 1]] for p, s in parts]).reshape(-1))
            n = sum(s[dst + 1] - s[dst] for _, s in parts)
            res.append(native.DeviceTable(outs[0].schema, n, vals, [None] * len(vals), "cuda:0"))
        return res

    def stage(name, inputs_per_rank):
        plan, ncols, key = st[name]
        return exchange_all([eng.run_device(plan, ins, ncols) for ins in inputs_per_rank], key)

    c = stage("customer", [[shard(customer, r)] for r in range(world)])
    o = stage("orders", [[shard(orders, r)] for r in range(world)])
    j1 = stage("join1", [[c[r], o[r]] for r in range(world)])
    l = stage("lineitem", [[shard(lineitem, r)] for r in range(world)])
    plan, ncols, _ = st["join2agg"]
    partials = [eng.run_host(plan, [j1[r], l[r]], ncols) for r in range(world)]
    return [p for p in partials if p is not None and p.num_rows]


def test_staged_q3_over_four_simulated_ranks_equals_oracle(built):
    from oracle import oracle as O
    customer, orders, lineitem = tpch.q3_tables(40_000, seed=7)
    partials = _simulated_ranks_q3(4, customer, orders, lineitem)
    assert len(partials) == 4                         # every partition received groups
    got = pa.concat_tables(partials)
    want = O.run_plan_to_arrow(S, tpch.q3_plan(), [customer, orders, lineitem])
    assert got.num_rows == want.num_rows > 100        # groups are partition-local: no group appears on two ranks
    assert _rows(got) == _rows(want)


def test_q3_distributed_on_one_rank_rccl_group(built):
    import os
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29950 + os.getpid() % 40))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        customer, orders, lineitem = tpch.q3_tables(30_000, seed=8)
        dt = lambda tb: native.DeviceTable.from_arrow(tb)
        timings = {}
        top, groups = parallel.run_q3_distributed(parallel.GpuEngine(), parallel.HipPartitioner(), dt(customer), dt(orders), dt(lineitem), timings=timings)
        plan = tpch.q3_plan()
        partial = O.run_plan_to_arrow(S, plan, [customer, orders, lineitem])
        final = O.run_plan_to_arrow(S, S.final_of(plan, partial.schema), partial)
        assert top == parallel.q3_top10(final) and len(top) == 10
        assert groups == final.num_rows
        assert timings["exchange_rows"] > 0
    finally:
        dist.destroy_process_group()
