"""Parity at BASELINE.json's FULL sizes, through size-independent properties (the oracle would take minutes there):
exact agreement with independent torch reductions of the generating tensors (int64 arithmetic, dense lookups instead of hash
tables), linearity (the Final of per-shard Partials equals the whole), and the reference's golden output shape.
SF10 Q6 = 59,986,052 rows, SF100 Q1 = 600,037,902 rows, SF10 Q3 = 15 M orders / 60 M lineitems (SF100 runs in tools/q3_dist.py)."""
import datetime

import pyarrow as pa
import pytest

from datafusion_comet_amd import native, parallel, serde as S, tpch

pytestmark = pytest.mark.gpu


def _run_device(plan, dt, ncols):
    return pa.Table.from_batches(native.execute_to_table([native.DeviceInput(dt)], ncols, plan.encode(), batch_size=0))


def _slice(dt: native.DeviceTable, start: int, length: int) -> native.DeviceTable:
    vals = []
    for f, v in zip(dt.schema, dt.values):
        w = native.value_width(f.type)
        vals.append(v[start * w:(start + length) * w])
    return native.DeviceTable(dt.schema, length, vals, [None] * len(vals), dt.device)


def test_q6_sf10_exact_and_linear(built):
    import torch
    n = 59_986_052
    dt, chk = tpch.lineitem_q6_device(n)
    want = tpch.q6_torch_reference(chk)
    plan = tpch.q6_plan()
    whole = _run_device(plan, dt, tpch.Q6_NUM_OUTPUT_COLS)
    assert int(whole.column(0)[0].as_py().scaleb(4)) == want and whole.column(1)[0].as_py() is False
    # linearity: 5 uneven row-range shards → Partial each → Final == whole
    cuts = [0, 1, 7_000_003, 20_000_000, 59_000_000, n]
    states = pa.concat_tables([_run_device(plan, _slice(dt, cuts[i], cuts[i + 1] - cuts[i]), tpch.Q6_NUM_OUTPUT_COLS) for i in range(5)])
    fin = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(states)], 1, S.final_of(plan, states.schema).encode()))
    assert int(fin.column(0)[0].as_py().scaleb(4)) == want
    assert fin.schema.field(0).type == pa.decimal128(35, 4)      # golden schema of q6.sql.out
    del dt, chk
    torch.cuda.empty_cache()


def test_q1_sf100_groups_match_torch_reductions(built):
    import torch
    n = 600_037_902
    dt, chk = tpch.lineitem_q1_device(n)
    torch.cuda.synchronize()
    out = _run_device(tpch.q1_plan(), dt, tpch.Q1_NUM_OUTPUT_COLS)
    assert out.num_rows == 4                                      # A/F, N/F, N/O, R/F like dbgen (q1.sql.out:6-9)
    assert tpch.q1_check_against_torch(out, chk) == []
    assert sum(out.column(out.num_columns - 1).to_pylist()) == int((chk["ship"] <= tpch.days(1998, 9, 2)).sum().item())
    del dt, chk
    torch.cuda.empty_cache()


def test_q3_sf10_staged_with_exchanges_matches_torch(built):
    import torch
    n_orders = 15_000_000
    customer, orders, lineitem, _ = tpch.q3_tables_device(n_orders, 1, 0, "cuda:0", 3)
    top, groups = parallel.run_q3_distributed(parallel.GpuEngine(0), parallel.HipPartitioner(), customer, orders, lineitem)
    del customer, orders, lineitem
    torch.cuda.empty_cache()
    want, want_groups = tpch.q3_torch_reference(n_orders, 1, "cuda:0", 3)
    got = [(r[0], (r[1] - datetime.date(1970, 1, 1)).days, r[2], int(r[3].scaleb(4))) for r in top]
    assert got == want and groups == want_groups and len(got) == 10
    torch.cuda.empty_cache()
