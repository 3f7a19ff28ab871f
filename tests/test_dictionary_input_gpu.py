"""Dictionary-encoded inputs are unpacked at the scan (operators/scan.rs:98-106, copy.rs:69-93) — here by a device
gather kernel.  First case is the reference's own planner test: dictionary Int32 keys n % 4, filter `col = 3`,
100 rows → 25 rows (planner.rs:4637-4699 test_unpack_dictionary_primitive)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu


def _run(plan, table, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(table, **({"batch_rows": kw.pop("batch_rows")} if "batch_rows" in kw else {}))],
                                  ncols, plan.encode(), **kw)
    return pa.Table.from_batches(out) if out else None


def test_reference_unpack_dictionary_primitive(built):
    values = pa.array([0, 1, 2, 3], pa.int32())
    keys = pa.array([i % 4 for i in range(100)], pa.int32())
    table = pa.table({"c": pa.DictionaryArray.from_arrays(keys, values)})
    plan = S.filter_(S.scan([S.T_INT32]), S.eq(S.col(0, S.T_INT32), S.lit(3, S.T_INT32)))
    got = _run(plan, table, 1)
    assert got.num_rows == 25
    assert got.column(0).type == pa.int32()          # downstream never sees a dictionary array
    assert got.column(0).to_pylist() == [3] * 25


def test_dictionary_with_nulls_and_many_batches_matches_plain(built):
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 50_000
    dict_vals = pa.array([10, None, -7, 2**40, 5], pa.int64())
    idx = pa.array(rng.integers(0, 5, n), pa.int16(), mask=rng.random(n) < 0.1)
    plain = pa.DictionaryArray.from_arrays(idx, dict_vals).dictionary_decode()
    other = pa.array(rng.integers(0, 100, n), pa.int32())
    t_dict = pa.table({"d": pa.DictionaryArray.from_arrays(idx, dict_vals), "o": other})
    t_plain = pa.table({"d": plain, "o": other})
    c0, c1 = S.col(0, S.T_INT64), S.col(1, S.T_INT32)
    plan = S.project(S.filter_(S.scan([S.T_INT64, S.T_INT32]), S.lt(c1, S.lit(50, S.T_INT32))), [c0, S.math("add", c0, S.lit(1, S.T_INT64), S.T_INT64)])
    got = _run(plan, t_dict, 2, batch_rows=3000)
    want = O.run_plan_to_arrow(S, plan, t_plain)
    assert got.column(0).combine_chunks().equals(want.column(0).combine_chunks())
    assert got.column(1).combine_chunks().equals(want.column(1).combine_chunks())


def test_dictionary_utf8_group_keys(built):
    from oracle import oracle as O
    rng = np.random.default_rng(6)
    n = 40_000
    words = pa.array(["AIR", "RAIL", None, "TRUCK", "SHIP", ""], pa.utf8())
    idx = pa.array(rng.integers(0, 6, n), pa.int32(), mask=rng.random(n) < 0.05)
    vals = pa.array(rng.integers(-1000, 1000, n), pa.int64())
    t_dict = pa.table({"k": pa.DictionaryArray.from_arrays(idx, words), "v": vals})
    t_plain = pa.table({"k": t_dict.column(0).combine_chunks().dictionary_decode(), "v": vals})
    plan = S.hash_agg(S.scan([S.T_STRING, S.T_INT64]), [S.col(0, S.T_STRING)], [S.sum_(S.col(1, S.T_INT64), S.T_INT64), S.count(S.lit(1, S.T_INT32))])
    got = _run(plan, t_dict, 3, batch_rows=4096)
    want = O.run_plan_to_arrow(S, plan, t_plain)
    key = lambda r: (r[0] is None, r[0] or "")
    assert sorted(zip(*[got.column(i).to_pylist() for i in range(3)]), key=key) == sorted(zip(*[want.column(i).to_pylist() for i in range(3)]), key=key)
    assert got.num_rows == 6   # 5 distinct non-null values (incl. "") + NULL
