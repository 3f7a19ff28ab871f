"""Columnar → UnsafeRow on the GPU (Native.columnarToRow*, columnar_to_row.rs): the engine's rows against the oracle's
restatement of the reference's writer, byte for byte — every supported type, NULLs, empty and multi-byte strings, wide decimals of every
byte length, > 64 columns (two bitset words), empty batches, and reuse of one converter for several batches."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native

pytestmark = pytest.mark.gpu


def _batch(n, seed=3):
    rng = np.random.default_rng(seed)
    m = lambda p: rng.random(n) < p
    wide = [None if i % 9 == 0 else decimal.Decimal(int(x)).scaleb(-4) for i, x in enumerate([(-1) ** i * (3 ** (i % 70)) for i in range(n)])]
    return pa.record_batch({
        "b": pa.array(rng.random(n) < 0.5, mask=m(0.1)), "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=m(0.1)),
        "i16": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16)), "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=m(0.2)),
        "i64": pa.array(rng.integers(-2**62, 2**62, n), mask=m(0.1)), "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m(0.1)),
        "f64": pa.array(np.where(rng.random(n) < 0.05, np.nan, rng.standard_normal(n)), mask=m(0.1)),
        "date": pa.array(rng.integers(-1000, 20000, n).astype(np.int32), pa.date32(), mask=m(0.1)), "ts": pa.array(rng.integers(-10**15, 10**15, n), pa.timestamp("us", tz="UTC")),
        "dec": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**11, 10**11, n)], pa.decimal128(12, 2)),
        "wide": pa.array(wide, pa.decimal128(38, 4)),
        "s": pa.array([None if i % 7 == 0 else ("" if i % 5 == 0 else "värde-%d" % i * (i % 4)) for i in range(n)], pa.string()),
        "bin": pa.array([None if i % 11 == 0 else bytes([i % 251]) * (i % 17) for i in range(n)], pa.binary()),
    })


def test_rows_match_the_reference_layout(built):
    from oracle import shuffle_oracle as SO
    c = native.ColumnarToRow()
    for n, seed in ((1, 1), (1000, 2), (20_000, 3)):
        b = _batch(n, seed)
        got, want = c.convert(b), SO.unsafe_rows(b)
        assert len(got) == len(want) == n
        assert got == want
    assert c.convert(_batch(10).slice(0, 0)) == []
    c.close()


def test_more_than_64_columns_and_partial_batches(built):
    from oracle import shuffle_oracle as SO
    rng = np.random.default_rng(5)
    n = 500
    b = pa.record_batch({"c%d" % i: pa.array(rng.integers(-100, 100, n), pa.int64(), mask=rng.random(n) < 0.3) if i % 3 else pa.array(["s%d" % (i * j % 13) for j in range(n)], pa.string())
                         for i in range(70)})
    c = native.ColumnarToRow()
    assert c.convert(b) == SO.unsafe_rows(b)
    assert c.convert(b, num_rows=123) == SO.unsafe_rows(b.slice(0, 123))        # numRows may be smaller than the arrays
    c.close()


def test_through_the_jni_exports(built):
    """Native.columnarToRowInit / Convert / Close driven through a JVM-less JNIEnv: the shim builds NativeColumnarToRowInfo(long, int[], int[])."""
    from oracle import shuffle_oracle as SO
    from tests.jni_mock import Jvm
    jvm = Jvm(native.lib())
    b = _batch(3000, 7)
    rows = jvm.columnar_to_row(b)
    assert rows is not None, jvm.exception()
    assert rows == SO.unsafe_rows(b)


def test_dictionary_encoded_and_sliced_inputs(built):
    """Dictionary arrays are unpacked first (the reference: columnar_to_row.rs casts dictionaries to their value type) — every index width,
    NULL indices and NULL dictionary entries, strings / decimals / booleans as values — and arrays exported with a non-zero offset (a
    slice of a larger batch: validity and Boolean bits start mid-byte) convert like their materialised copies."""
    from oracle import shuffle_oracle as SO
    rng = np.random.default_rng(8)
    n = 3000
    plain = _batch(n, 8)
    cols, names = [], []
    words = pa.array([None if i % 13 == 0 else ["", "a", "lineitem", "värde", "x" * 40][i % 5] for i in range(n)], pa.string())
    plain = plain.append_column("low", words)
    for name, idx_type in (("low", pa.int8()), ("s", pa.int32()), ("dec", pa.int16()), ("i64", pa.int32()), ("f64", pa.int64()), ("b", pa.uint8()), ("wide", pa.uint16()), ("date", pa.int32())):
        cols.append(plain.column(name).dictionary_encode().cast(pa.dictionary(idx_type, plain.column(name).type)))
        names.append(name)
    # a dictionary whose ENTRY is NULL (not only its indices)
    d = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 3, n), pa.int32(), mask=rng.random(n) < 0.1), pa.array(["x", None, "zzz"]))
    cols.append(d)
    names.append("dnull")
    b = pa.record_batch(cols, names=names)
    want_src = pa.record_batch([c.dictionary_decode() for c in cols], names=names)
    c = native.ColumnarToRow()
    assert c.convert(b) == SO.unsafe_rows(want_src)
    # slices: offsets 1, 7 and 1001 (not byte aligned), through plain and dictionary columns alike
    for off, length in ((1, 50), (7, 1000), (1001, 1999)):
        assert c.convert(plain.slice(off, length)) == SO.unsafe_rows(pa.record_batch([x.take(pa.array(range(off, off + length))) for x in plain.columns], names=plain.schema.names))
        assert c.convert(b.slice(off, length)) == SO.unsafe_rows(want_src.slice(off, length))
    c.close()


def test_nested_types(built):
    """struct / list / map columns (columnar_to_row.rs:570-830, 1602-1900): a nested row per struct, UnsafeArrayData per list (elements at their
    natural width, variable-length elements behind them), key and value arrays per map — nested in each other, NULL at every level, empty
    lists and maps, wide decimals and strings inside, next to flat columns; sliced input included.  These batches are written on the host."""
    from oracle import shuffle_oracle as SO
    rng = np.random.default_rng(12)
    n = 2000
    def maybe(p, f):
        return None if rng.random() < p else f()
    ints = lambda: [maybe(0.2, lambda: int(rng.integers(-2**31, 2**31))) for _ in range(int(rng.integers(0, 6)))]
    strs = lambda: [maybe(0.2, lambda: "é" * int(rng.integers(0, 12))) for _ in range(int(rng.integers(0, 4)))]
    point = lambda: {"x": maybe(0.1, lambda: float(rng.standard_normal())), "tag": maybe(0.2, lambda: "t%d" % int(rng.integers(0, 1000))),
                     "amt": maybe(0.2, lambda: decimal.Decimal(int(rng.integers(-10**9, 10**9)) * 10**20).scaleb(-4)), "pts": maybe(0.2, ints)}
    point_t = pa.struct([("x", pa.float64()), ("tag", pa.string()), ("amt", pa.decimal128(38, 4)), ("pts", pa.list_(pa.int32()))])
    b = pa.record_batch({
        "id": pa.array(np.arange(n, dtype=np.int64)),
        "ints": pa.array([maybe(0.1, ints) for _ in range(n)], pa.list_(pa.int32())),
        "strs": pa.array([maybe(0.1, strs) for _ in range(n)], pa.list_(pa.string())),
        "bools": pa.array([maybe(0.1, lambda: [maybe(0.2, lambda: bool(rng.random() < 0.5)) for _ in range(int(rng.integers(0, 70)))]) for _ in range(n)], pa.list_(pa.bool_())),
        "point": pa.array([maybe(0.1, point) for _ in range(n)], point_t),
        "points": pa.array([maybe(0.1, lambda: [maybe(0.2, point) for _ in range(int(rng.integers(0, 3)))]) for _ in range(n)], pa.list_(point_t)),
        "m": pa.array([maybe(0.1, lambda: [("k%d" % j, maybe(0.2, lambda: int(rng.integers(0, 10**12)))) for j in range(int(rng.integers(0, 4)))]) for _ in range(n)],
                      pa.map_(pa.string(), pa.int64())),
        "mm": pa.array([maybe(0.2, lambda: [(int(j), [("a", 1.5), ("b", None)][: int(rng.integers(0, 3))]) for j in range(int(rng.integers(0, 3)))]) for _ in range(n)],
                       pa.map_(pa.int32(), pa.map_(pa.string(), pa.float64()))),
        "ll": pa.array([maybe(0.1, lambda: [maybe(0.2, ints) for _ in range(int(rng.integers(0, 3)))]) for _ in range(n)], pa.list_(pa.list_(pa.int32()))),
        "dec": pa.array([maybe(0.1, lambda: [decimal.Decimal(int(rng.integers(-10**10, 10**10))).scaleb(-2)]) for _ in range(n)], pa.list_(pa.decimal128(12, 2))),
        "s": pa.array(["row %d" % i for i in range(n)], pa.string()),
    })
    c = native.ColumnarToRow()
    got, want = c.convert(b), SO.unsafe_rows(b)
    assert len(got) == n
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"row {i}"
    sl = b.slice(777, 1000)
    assert c.convert(sl) == SO.unsafe_rows(pa.record_batch([x.take(pa.array(range(777, 1777))) for x in b.columns], names=b.schema.names))
    c.close()
