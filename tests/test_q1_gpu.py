"""GPU parity: TPC-H Q1 stage 1 — filter + decimal projection (incl. the 256-bit wide-decimal path) + partial
hash aggregate with two Utf8 keys and eight aggregates.  Group order is unspecified in the reference
(SURVEY §7), so rows are compared as a multiset keyed by the group keys; every state column is bit-exact."""
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _rows(tbl: pa.Table):
    cols = [tbl.column(i).to_pylist() for i in range(tbl.num_columns)]
    return sorted(zip(*cols), key=lambda r: tuple("" if x is None else str(x) for x in r[:2]))


def _oracle(plan, table):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, table)


def _run(plan, inputs, ncols, **kw):
    out = native.execute_to_table(inputs, ncols, plan.encode(), **kw)
    return pa.Table.from_batches(out) if out else None


@pytest.mark.parametrize("n", [1, 100, 8192, 200_003])
def test_q1_host_stream_matches_oracle(built, n):
    table = tpch.lineitem_q1(n, seed=n)
    plan = tpch.q1_plan()
    got = _run(plan, [native.HostInput.from_table(table)], tpch.Q1_NUM_OUTPUT_COLS)
    want = _oracle(plan, table)
    assert got.num_columns == tpch.Q1_NUM_OUTPUT_COLS
    assert _rows(got) == _rows(want)
    assert got.schema.field(8).type == pa.decimal128(38, 6)   # sum_charge state
    assert got.schema.field(0).type == pa.utf8()


def test_q1_device_resident_matches_oracle(built):
    table = tpch.lineitem_q1(1_500_000, seed=7)
    plan = tpch.q1_plan()
    dev = native.DeviceTable.from_arrow(table, "cuda:0")
    got = _run(plan, [native.DeviceInput(dev)], tpch.Q1_NUM_OUTPUT_COLS)
    want = _oracle(plan, table)
    assert got.num_rows == 4
    assert _rows(got) == _rows(want)


def test_q1_device_resident_with_fixed_length_metadata(built):
    """The owner of a resident table may declare a Utf8 column's uniform value length in the field metadata (comet:utf8_fixed_len, measured
    once by DeviceTable.with_string_hints); the engine then checks the end points only.  Same answer as without the metadata; a declaration
    the batch's end points contradict is refused; columns whose lengths vary get no metadata."""
    table = tpch.lineitem_q1(300_000, seed=11)
    plan = tpch.q1_plan()
    dev = native.DeviceTable.from_arrow(table, "cuda:0")
    hinted = dev.with_string_hints()
    flagged = [f.name for f in hinted.schema if f.metadata and b"comet:utf8_fixed_len" in f.metadata]
    assert len(flagged) == 2 and all(hinted.schema.field(n).metadata[b"comet:utf8_fixed_len"] == b"1" for n in flagged)
    got = _run(plan, [native.DeviceInput(hinted)], tpch.Q1_NUM_OUTPUT_COLS)
    assert _rows(got) == _rows(_oracle(plan, table))
    # a wrong declaration (2 bytes per value over one-byte values) does not survive the end-point check
    lying = native.DeviceTable(pa.schema([f.with_metadata({b"comet:utf8_fixed_len": b"2"}) if f.name in flagged else f for f in dev.schema]),
                               dev.num_rows, dev.values, dev.validity, dev.device, dev.aux)
    with pytest.raises(native.CometNativeException, match="comet:utf8_fixed_len=2"):
        _run(plan, [native.DeviceInput(lying)], tpch.Q1_NUM_OUTPUT_COLS)
    # values of different lengths: nothing is declared
    mixed = native.DeviceTable.from_arrow(pa.table({"s": pa.array(["a", "bc", "d", ""])}), "cuda:0").with_string_hints()
    assert not mixed.schema.field(0).metadata


def test_fixed_length_declaration_that_keeps_the_byte_total_is_refused(built):
    """lengths 0,2,0,2,… declared as 1 keep the byte total (and both end points) right.  Every declared column is
    verified by utf8_uniform_kernel the first time its buffer is seen (the verdict is cached per buffer address afterwards), so the lie is
    refused instead of producing wrong groups; a truthful declaration on the same plan still runs, twice (second task = cached verdict)."""
    n = 4096
    keys = ["" if i % 2 == 0 else "ab" for i in range(n)]
    t = pa.table({"k": pa.array(keys, pa.string()), "v": pa.array(range(n), pa.int64())})
    plan = S.hash_agg(S.scan([S.T_STRING, S.T_INT64]), [S.col(0, S.T_STRING)], [S.count(S.col(1, S.T_INT64))], mode=S.PARTIAL)
    dev = native.DeviceTable.from_arrow(t, "cuda:0")
    assert _rows(_run(plan, [native.DeviceInput(dev)], 2)) == [("", n // 2), ("ab", n // 2)]
    lying = native.DeviceTable(pa.schema([dev.schema.field(0).with_metadata({b"comet:utf8_fixed_len": b"1"}), dev.schema.field(1)]),
                               dev.num_rows, dev.values, dev.validity, dev.device, dev.aux)
    for _ in range(2):          # never cached as good
        with pytest.raises(native.CometNativeException, match="offsets are not 1 bytes apart"):
            _run(plan, [native.DeviceInput(lying)], 2)
    t2 = pa.table({"k": pa.array(["x" if i % 3 else "y" for i in range(n)], pa.string()), "v": pa.array(range(n), pa.int64())})
    dev2 = native.DeviceTable.from_arrow(t2, "cuda:0").with_string_hints()
    assert dev2.schema.field(0).metadata[b"comet:utf8_fixed_len"] == b"1"
    a = _rows(_run(plan, [native.DeviceInput(dev2)], 2))
    b = _rows(_run(plan, [native.DeviceInput(dev2)], 2))
    assert a == b == [("x", n - (n + 2) // 3), ("y", (n + 2) // 3)]


def test_q1_chunked_equals_unchunked(built):
    table = tpch.lineitem_q1(100_000, seed=2)
    plan = tpch.q1_plan()
    cfg = S.config_map({"spark.comet.gpu.chunkRows": 7_000})
    a = _run(plan, [native.HostInput.from_table(table, batch_rows=1000)], tpch.Q1_NUM_OUTPUT_COLS, config=cfg)
    b = _run(plan, [native.HostInput.from_table(table)], tpch.Q1_NUM_OUTPUT_COLS)
    assert _rows(a) == _rows(b)


def test_grouped_empty_input_emits_nothing(built):
    table = tpch.lineitem_q1(0)
    out = native.execute_to_table([native.HostInput.from_table(table)], tpch.Q1_NUM_OUTPUT_COLS, tpch.q1_plan().encode())
    assert out == []


def test_high_cardinality_int_keys_grow_the_table(built):
    # 300k distinct int64 keys: exceeds the LDS table and the initial 64k-slot global table (growth + rehash path)
    import numpy as np
    n = 600_000
    rng = np.random.default_rng(9)
    keys = rng.integers(0, 300_000, n, dtype=np.int64)
    vals = rng.integers(-1000, 1000, n, dtype=np.int64)
    table = pa.table({"k": pa.array(keys), "v": pa.array(vals)})
    plan = S.hash_agg(S.scan([S.T_INT64, S.T_INT64]), [S.col(0, S.T_INT64)],
                      [S.sum_(S.col(1, S.T_INT64), S.T_INT64), S.count(S.col(1, S.T_INT64)), S.min_(S.col(1, S.T_INT64), S.T_INT64),
                       S.max_(S.col(1, S.T_INT64), S.T_INT64)])
    got = _run(plan, [native.HostInput.from_table(table)], 5, batch_size=0)
    want = _oracle(plan, table)
    g = sorted(zip(*[got.column(i).to_pylist() for i in range(5)]))
    w = sorted(zip(*[want.column(i).to_pylist() for i in range(5)]))
    assert len(g) == len(w) == len(set(keys.tolist()))
    assert g == w


def test_null_group_keys_and_null_values(built):
    import numpy as np
    n = 50_000
    rng = np.random.default_rng(4)
    k = pa.array(rng.integers(0, 7, n), pa.int32(), mask=rng.random(n) < 0.1)
    v = pa.array(rng.integers(-10**9, 10**9, n), pa.int64(), mask=rng.random(n) < 0.2)
    table = pa.table({"k": k, "v": v})
    plan = S.hash_agg(S.scan([S.T_INT32, S.T_INT64]), [S.col(0, S.T_INT32)],
                      [S.sum_(S.col(1, S.T_INT64), S.T_INT64), S.count(S.col(1, S.T_INT64)), S.count(S.lit(1, S.T_INT32))])
    got = _run(plan, [native.HostInput.from_table(table)], 4)
    want = _oracle(plan, table)
    key = lambda r: (r[0] is None, r[0] or 0)
    assert sorted(zip(*[got.column(i).to_pylist() for i in range(4)]), key=key) == \
        sorted(zip(*[want.column(i).to_pylist() for i in range(4)]), key=key)
