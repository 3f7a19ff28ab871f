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
