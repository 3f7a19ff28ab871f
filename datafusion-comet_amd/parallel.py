"""Multi-GPU execution of a scan→filter→aggregate plan: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in CPU tests), rows sharded by contiguous ranges — the unit Spark already uses (one native
plan per partition, jni_api.rs:823-825) — so the data path needs NO collective.  Only the Partial
state rows (≤ one row per group per rank) are gathered to rank 0, which runs the plan's Final stage.
"""
from __future__ import annotations

import io
from typing import Callable, Optional, Tuple

import pyarrow as pa


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [start, start+length) of `rank`; ranges tile [0, n_rows) exactly, sizes differ by ≤ 1."""
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def _to_ipc(table: Optional[pa.Table]) -> bytes:
    if table is None:
        return b""
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, table.schema) as w:
        w.write_table(table)
    return sink.getvalue()


def _from_ipc(b: bytes) -> Optional[pa.Table]:
    if not b:
        return None
    return pa.ipc.open_stream(io.BytesIO(b)).read_all()


def gather_partial_states(local_states: Optional[pa.Table], dst: int = 0, group=None) -> Optional[pa.Table]:
    """Gather every rank's Partial-state table on `dst` (tiny: rows = groups per rank).  Returns the concatenation on
    `dst`, None elsewhere.  Without an initialised process group this is the identity (single GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_states
    rank = dist.get_rank(group)
    payload = _to_ipc(local_states)
    out = [None] * dist.get_world_size(group) if rank == dst else None
    dist.gather_object(payload, out, dst=dst, group=group)
    if rank != dst:
        return None
    tables = [t for t in (_from_ipc(b) for b in out) if t is not None and t.num_rows > 0]
    if not tables:
        return _from_ipc(out[0]) if out and out[0] else None
    return pa.concat_tables(tables)


def run_sharded_aggregate(table_rows: int, load_shard: Callable[[int, int], pa.Table], run_partial: Callable[[pa.Table], Optional[pa.Table]],
                          run_final: Callable[[pa.Table], pa.Table], group=None) -> Optional[pa.Table]:
    """rank r: load rows shard_range(r) → Partial plan → gather states on rank 0 → Final plan there."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, length = shard_range(table_rows, world, rank)
    states = run_partial(load_shard(start, length))
    gathered = gather_partial_states(states, 0, group)
    if rank != 0 or gathered is None:
        return None
    return run_final(gathered)


# ------------------------------------------------------------------------------------------------------------------
# Exchange between stages (hash-partitioned plans: joins, high-cardinality aggregates) — SURVEY §8e / config 4.
#
# A stage's output stays in HBM (comet_execute_plan_device); rows are assigned to partitions exactly like Spark's
# HashPartitioning (murmur3 seed 42 chained over the key columns, pmod num_partitions — the reference's shuffle writer,
# native/shuffle/src/partitioners/multi_partition.rs:280-330), grouped by partition on device (comet_partition_indices +
# comet_take_column) and moved with ONE all-to-all per buffer over RCCL (torch.distributed backend "nccl"; xGMI is
# point-to-point, so the all-to-all maps 1:1 onto the links).  partition p of the exchange lives on rank p.
# ------------------------------------------------------------------------------------------------------------------


class HipPartitioner:
    """The product partitioner: every step is a HIP kernel of libcomet.so on the table's GPU."""

    def __call__(self, table, key_cols, num_partitions):
        from . import native
        pids = native.partition_ids(table, key_cols, num_partitions)
        return native.partition_table(table, pids, num_partitions)


class NativeExchange:
    """Exchange done inside libcomet.so (csrc/exchange.cpp): partitioning kernels + RCCL send/recv groups (or the in-process transport),
    no torch collective on the data path.  `comm` is a native.NativeComm."""

    def __init__(self, comm):
        self.comm = comm


def native_comm_from_process_group(device_id: int, group=None):
    """One NativeComm per rank of the torch process group: rank 0 draws the RCCL unique id, the launcher's rendezvous (the control
    plane only) carries it to the other ranks."""
    import torch.distributed as dist
    from . import native
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [native.NativeComm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return native.NativeComm(world, rank, device_id, unique_id=box[0])


def _unpack_bits(bits, n):
    import torch
    sh = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((bits[:, None] >> sh) & 1).reshape(-1)[:n].contiguous()


def _pack_bits(bytes_):
    import torch
    n = bytes_.numel()
    pad = (-n) % 8
    if pad:
        bytes_ = torch.cat([bytes_, torch.zeros(pad, dtype=torch.uint8, device=bytes_.device)])
    sh = torch.arange(8, device=bytes_.device, dtype=torch.uint8)
    return (bytes_.reshape(-1, 8) << sh).sum(dim=1, dtype=torch.int32).to(torch.uint8)


def exchange(table, key_cols, partitioner, group=None):
    """Hash-repartition `table` (a native.DeviceTable; this rank's part of a distributed table) on `key_cols` across the
    ranks of `group`.  Returns the rows whose partition id equals this rank, as a DeviceTable.  Rows from one sender keep
    their order; senders are concatenated in rank order."""
    import pyarrow as pa
    import torch
    import torch.distributed as dist
    from .native import DeviceTable, value_width
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return table        # one partition: nothing to hash or move (the reference's writer takes its SinglePartition path, no hashing)
    if isinstance(partitioner, NativeExchange):
        return partitioner.comm.exchange(table, key_cols)
    part, starts = partitioner(table, key_cols, world)
    dev = part.values[0].device if part.values else torch.device(table.device)
    send = [starts[i + 1] - starts[i] for i in range(world)]
    t_send = torch.tensor(send, dtype=torch.int64, device=dev)
    t_recv = torch.empty_like(t_send)
    dist.all_to_all_single(t_recv, t_send, group=group)
    recv = [int(x) for x in t_recv.tolist()]
    n_in, n_out = part.num_rows, sum(recv)
    # a column carries validity after the exchange iff it does on any rank
    flags = torch.tensor([1 if v is not None else 0 for v in part.validity], dtype=torch.int32, device=dev)
    if flags.numel():
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    any_valid = [bool(x) for x in flags.tolist()]

    def a2a(rows):   # rows: [n_in, w] uint8
        out = torch.empty((n_out, rows.shape[1]), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=recv, input_split_sizes=send, group=group)
        return out

    vals, valid, auxs = [], [], []
    for i, f in enumerate(part.schema):
        if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
            # Utf8: the lengths travel one int32 per row, the bytes with per-partition BYTE split sizes; the receiver rebuilds
            # the offsets with a prefix sum
            offs = part.values[i].view(torch.int32)[: n_in + 1]
            lens = (offs[1:] - offs[:-1]).contiguous()
            cut = offs[torch.tensor(starts, dtype=torch.int64, device=dev)].to(torch.int64)
            send_b = (cut[1:] - cut[:-1]).contiguous()
            recv_b = torch.empty_like(send_b)
            dist.all_to_all_single(recv_b, send_b, group=group)
            send_bl, recv_bl = [int(x) for x in send_b.tolist()], [int(x) for x in recv_b.tolist()]
            rl = a2a(lens.view(torch.uint8).reshape(n_in, 4)).reshape(-1).view(torch.int32)
            data_in = part.aux[i][: sum(send_bl)].reshape(-1, 1)
            data_out = torch.empty((sum(recv_bl), 1), dtype=torch.uint8, device=dev)
            dist.all_to_all_single(data_out, data_in.contiguous(), output_split_sizes=recv_bl, input_split_sizes=send_bl, group=group)
            new_offs = torch.zeros(n_out + 1, dtype=torch.int32, device=dev)
            if n_out:
                new_offs[1:] = torch.cumsum(rl, 0, dtype=torch.int32)
            vals.append(new_offs.view(torch.uint8).reshape(-1))
            auxs.append(data_out.reshape(-1) if data_out.numel() else torch.zeros(1, dtype=torch.uint8, device=dev))
            if any_valid[i]:
                vb = _unpack_bits(part.validity[i], n_in) if part.validity[i] is not None else torch.ones(n_in, dtype=torch.uint8, device=dev)
                valid.append(_pack_bits(a2a(vb.reshape(n_in, 1)).reshape(-1)))
            else:
                valid.append(None)
            continue
        auxs.append(None)
        w = value_width(f.type)
        if w == 0:   # Boolean values travel one byte per row (partition boundaries are not byte aligned)
            vals.append(_pack_bits(a2a(_unpack_bits(part.values[i], n_in).reshape(n_in, 1)).reshape(-1)))
        else:
            vals.append(a2a(part.values[i].reshape(n_in, w)).reshape(-1))
        if any_valid[i]:
            vb = _unpack_bits(part.validity[i], n_in) if part.validity[i] is not None else torch.ones(n_in, dtype=torch.uint8, device=dev)
            valid.append(_pack_bits(a2a(vb.reshape(n_in, 1)).reshape(-1)))
        else:
            valid.append(None)
    if dev.type == "cuda":
        # the collectives were enqueued on torch's stream; libcomet reads these buffers on ITS OWN stream next
        torch.cuda.current_stream(dev).synchronize()
    return DeviceTable(part.schema, n_out, vals, valid, table.device, auxs)


class GpuEngine:
    """Runs one stage plan on this rank's GPU through the C ABI (device-resident inputs and outputs)."""

    def __init__(self, device_id: int = 0):
        self.device_id = device_id
        self.kernel_ms = 0.0

    def _inputs(self, tables):
        from . import native
        return [native.DeviceInput(t, self.device_id) if isinstance(t, native.DeviceTable) else native.HostInput.from_table(t) for t in tables]

    @staticmethod
    def _close(inputs):
        for i in inputs:
            if hasattr(i, "close"):
                i.close()

    def run_device(self, plan, tables, ncols):
        from . import native
        ins = self._inputs(tables)
        try:
            return native.execute_to_device(ins, ncols, plan if isinstance(plan, (bytes, bytearray)) else plan.encode(), device_id=self.device_id)
        finally:
            self._close(ins)

    def run_host(self, plan, tables, ncols):
        from . import native
        ins = self._inputs(tables)
        try:
            out = native.execute_to_table(ins, ncols, plan if isinstance(plan, (bytes, bytearray)) else plan.encode(), batch_size=0, device_id=self.device_id)
        finally:
            self._close(ins)
        return pa.Table.from_batches(out) if out else None


def q3_top10(final: Optional[pa.Table]) -> list:
    """ORDER BY revenue DESC, o_orderdate LIMIT 10 over (l_orderkey, o_orderdate, o_shippriority, revenue) rows — Spark's
    TakeOrderedAndProject, which stays on the JVM side (outside the native hot path)."""
    if final is None or final.num_rows == 0:
        return []
    import pyarrow.compute as pc
    t = final.rename_columns([f"c{i}" for i in range(final.num_columns)]).combine_chunks()
    k = pc.select_k_unstable(t, min(10, t.num_rows), [("c3", "descending"), ("c1", "ascending"), ("c0", "ascending")])
    t = t.take(k)
    return list(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]))


def run_q3_distributed(engine, partitioner, customer, orders, lineitem, group=None, timings: Optional[dict] = None):
    """TPC-H Q3 over the ranks of `group`: each rank holds an arbitrary shard of the three tables.  Stages follow
    tpch.q3_stage_plans(); three exchanges on the join keys; aggregation is partition-local because l_orderkey is both the
    last exchange key and a group key.  Returns (top-10 rows on rank 0 else None, number of result groups on this rank)."""
    import time
    import torch.distributed as dist
    from . import serde as S, tpch
    st = tpch.q3_stage_plans()

    def clock():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        return time.perf_counter()

    def stage(name, inputs):
        plan, ncols, key = st[name]
        t0 = clock()
        out = engine.run_device(plan, inputs, ncols)
        t1 = clock()
        res = exchange(out, [key], partitioner, group)
        t2 = clock()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            timings["exchange"] = timings.get("exchange", 0.0) + (t2 - t1)
            timings["exchange_rows"] = timings.get("exchange_rows", 0) + out.num_rows
            timings["exchange_bytes"] = timings.get("exchange_bytes", 0) + out.nbytes()
        return res

    c = stage("customer", [customer])
    o = stage("orders", [orders])
    j1 = stage("join1", [c, o])
    l = stage("lineitem", [lineitem])
    plan, ncols, _ = st["join2agg"]
    t0 = clock()
    partial = engine.run_device(plan, [j1, l], ncols)      # Partial states stay in HBM
    t1 = clock()
    # Final aggregate + TakeOrdered (Sort with fetch 10: revenue DESC, o_orderdate ASC, then l_orderkey so that ties are
    # deterministic) in ONE native plan: only this rank's ten best rows leave the GPU
    groups = partial.num_rows                      # groups are partition-local: one Partial state row per group
    local = []
    if groups:
        f = S.final_of(plan, partial.schema)
        top = S.sort(f, [(S.col(3, S.decimal(36, 4)), True, True), (S.col(1, S.T_DATE), False, False), (S.col(0, S.T_INT64), False, False)], fetch=10)
        t = engine.run_host(top, [partial], 4)
        local = list(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)])) if t is not None else []
    if timings is not None:
        timings["join2agg"] = timings.get("join2agg", 0.0) + (t1 - t0)
        timings["final_agg_top10"] = timings.get("final_agg_top10", 0.0) + (clock() - t1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local, groups
    rank = dist.get_rank(group)
    out = [None] * dist.get_world_size(group) if rank == 0 else None
    dist.gather_object(local, out, dst=0, group=group)
    if rank != 0:
        return None, groups
    merged = [r for part in out for r in part]
    merged.sort(key=lambda r: (-r[3], r[1], r[0]))
    return merged[:10], groups


_Q95_PLANS = []


def _q95_plan_bytes(engine):
    """(stage A, stage B, leaves): plan bytes for an engine that takes bytes (GpuEngine), the trees for one that interprets them (the oracle stand-in of the CPU tests)"""
    from . import tpcds
    if not _Q95_PLANS:
        a, b, leaves = tpcds.q95_plans()
        _Q95_PLANS.append((a, b, leaves, a.encode(), b.encode()))
    a, b, leaves, ab, bb = _Q95_PLANS[0]
    return (ab, bb, leaves) if isinstance(engine, GpuEngine) else (a, b, leaves)


_Q3_SINGLE = []
_Q3_TOP_PLANS: dict = {}


def _q3_single_plan():
    if not _Q3_SINGLE:
        from . import tpch
        plan = tpch.q3_plan()
        _Q3_SINGLE.append((plan, plan.encode()))
    return _Q3_SINGLE[0]


def run_q3_single(engine, customer, orders, lineitem, timings: Optional[dict] = None):
    """TPC-H Q3 on ONE partition as ONE native plan (tpch.q3_plan: customer ⋈ orders ⋈ lineitem → Project → Partial aggregate) followed by
    the Final aggregate + TakeOrdered plan.  Nothing is exchanged, so nothing has to be materialised for an exchange: both probe sides are
    Filter / Projection chains over their scans and run INSIDE the probe kernels (JoinFusion) — the orders and lineitem tables are read
    once, by the probes.  Returns (top-10 rows, number of result groups)."""
    import time
    from . import serde as S, tpch

    def clock():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        return time.perf_counter()

    # the two plans' bytes are built once per process, like the serialised plan a Spark stage hands to every one of its tasks (CometExecIterator gets `protobufQueryPlan`
    # bytes, not a tree) — encoding the tree in Python per run cost the Final stage a quarter of its time
    plan, plan_bytes = _q3_single_plan()
    t0 = clock()
    partial = engine.run_device(plan_bytes, [customer, orders, lineitem], tpch.Q3_NUM_OUTPUT_COLS)
    t1 = clock()
    groups = partial.num_rows
    local = []
    if groups:
        key = str(partial.schema)
        top_bytes = _Q3_TOP_PLANS.get(key)
        if top_bytes is None:
            f = S.final_of(plan, partial.schema)
            top = S.sort(f, [(S.col(3, S.decimal(36, 4)), True, True), (S.col(1, S.T_DATE), False, False), (S.col(0, S.T_INT64), False, False)], fetch=10)
            top_bytes = _Q3_TOP_PLANS[key] = top.encode()
        t = engine.run_host(top_bytes, [partial], 4)
        local = list(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)])) if t is not None else []
    if timings is not None:
        timings["joins_partial_agg"] = timings.get("joins_partial_agg", 0.0) + (t1 - t0)
        timings["final_agg_top10"] = timings.get("final_agg_top10", 0.0) + (clock() - t1)
    return local, groups


def run_q95_distributed(engine, partitioner, tables, group=None, timings: Optional[dict] = None):
    """TPC-DS Q95 (BASELINE config 5) over the ranks of `group`.  `tables`: this rank's arbitrary shard of web_sales and web_returns plus
    full copies of date_dim / customer_address / web_site (the reference broadcasts those: BroadcastHashJoin in the approved plan).

    Every join of the fact tables — the ws_wh self-join, both LeftSemi joins, web_returns ⋈ ws_wh — is on the order number, which is also
    the count(DISTINCT) key, so ONE hash exchange of web_sales and one of web_returns on that key (the hashpartitioning(ws_order_number)
    exchanges under the plan's sort-merge joins) make everything up to the mixed-mode aggregate partition-local: each rank runs stage A
    of tpcds.q95_plans() on its partition and emits one row of states; the distinct counts add up because no order lives on two ranks.
    The state rows meet on rank 0 (Spark's single-partition exchange before the Final aggregate), which runs stage B.
    Returns (count, sum_cost, sum_profit) on rank 0, None elsewhere."""
    import time
    import torch.distributed as dist
    from . import native, tpcds
    # (the plans and their bytes once per process: a Spark stage serialises its plan once for all its tasks — encoding Q95's nine-leaf tree in Python per run
    # cost the bench leg 1.5 ms of its 16)
    stage_a, stage_b, leaves = _q95_plan_bytes(engine)

    def clock():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        return time.perf_counter()

    t0 = clock()
    ws = exchange(tables["web_sales"], [0], partitioner, group)
    wr = exchange(tables["web_returns"], [0], partitioner, group)
    t1 = clock()
    local = dict(tables, web_sales=ws, web_returns=wr)
    states = engine.run_host(stage_a, [local[n] for n in leaves], 5)      # one row: (sum, isEmpty, sum, isEmpty, count)
    t2 = clock()
    if timings is not None:
        timings["exchange"] = timings.get("exchange", 0.0) + (t1 - t0)
        timings["stage_a"] = timings.get("stage_a", 0.0) + (t2 - t1)
        timings["exchange_rows"] = timings.get("exchange_rows", 0) + tables["web_sales"].num_rows + tables["web_returns"].num_rows
        nb = lambda t: t.nbytes() if isinstance(t, native.DeviceTable) else t.nbytes
        timings["exchange_bytes"] = timings.get("exchange_bytes", 0) + nb(tables["web_sales"]) + nb(tables["web_returns"])
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    merged = gather_partial_states(states, 0, group) if distributed else states
    if distributed and dist.get_rank(group) != 0:
        return None
    out = engine.run_host(stage_b, [merged], 3)
    if timings is not None:
        timings["stage_b"] = timings.get("stage_b", 0.0) + (clock() - t2)
    return out.column(2)[0].as_py(), out.column(0)[0].as_py(), out.column(1)[0].as_py()
