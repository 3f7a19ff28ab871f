"""Nested columns through the Parquet scan and the plan above it (the reference: parquet/parquet_support.rs:166-383
parquet_convert_array / parquet_convert_struct_to_struct, GetStructField planner.rs:776-779): struct-of-flat and list-of-flat columns decoded
on the device from their leaves' definition / repetition levels, exported as Arrow struct / list arrays, passed through Filter / Projection
(gathered by row index, children and all), taken apart by GetStructField — every result against pyarrow's own reading of the same file."""
import datetime
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu


def _nested_table(n, seed, nullable=True):
    rng = np.random.default_rng(seed)

    def mask(p):
        return rng.random(n) < p if nullable else np.zeros(n, dtype=bool)

    a = rng.integers(-1000, 1000, n).astype(np.int32)
    b = ["s%d" % (v % 37) if v % 5 else "a much longer string value %d" % v for v in rng.integers(0, 10**6, n)]
    c = rng.standard_normal(n)
    d = [Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)]
    e = [datetime.date(1995, 1, 1) + datetime.timedelta(days=int(v)) for v in rng.integers(0, 3000, n)]
    ma, mb, mc, md, me, ms = mask(0.1), mask(0.15), mask(0.05), mask(0.2), mask(0.1), mask(0.12)
    st_type = pa.struct([pa.field("a", pa.int32(), nullable), pa.field("b", pa.string(), nullable), pa.field("c", pa.float64(), nullable),
                         pa.field("d", pa.decimal128(12, 2), nullable), pa.field("e", pa.date32(), nullable)])
    structs = []
    for i in range(n):
        if ms[i]:
            structs.append(None)
        else:
            structs.append({"a": None if ma[i] else int(a[i]), "b": None if mb[i] else b[i], "c": None if mc[i] else float(c[i]), "d": None if md[i] else d[i],
                            "e": None if me[i] else e[i]})
    lens = rng.integers(0, 9, n)
    lens[rng.random(n) < 0.1] = 0                                      # empty lists
    lens[rng.random(n) < 0.02] = 300                                   # a few long ones (they cross pages)
    lnull = mask(0.1)
    li, lf, ld = [], [], []
    for i in range(n):
        k = int(lens[i])
        if lnull[i]:
            li.append(None); lf.append(None); ld.append(None)
            continue
        vi = rng.integers(-10**12, 10**12, k)
        en = rng.random(k) < (0.15 if nullable else 0.0)
        li.append([None if en[j] else int(vi[j]) for j in range(k)])
        lf.append([None if en[j] else float(vi[j]) * 0.5 for j in range(k)])
        ld.append([None if en[j] else datetime.date(2000, 1, 1) + datetime.timedelta(days=int(vi[j] % 5000)) for j in range(k)])
    return pa.table({
        "k": pa.array(np.arange(n, dtype=np.int64)),
        "s": pa.array(structs, st_type),
        "li": pa.array(li, pa.list_(pa.field("element", pa.int64(), nullable))),
        "f": pa.array(rng.integers(0, 100, n).astype(np.int32), pa.int32(), mask=mask(0.1)),
        "lf": pa.array(lf, pa.list_(pa.field("element", pa.float64(), nullable))),
        "ld": pa.array(ld, pa.list_(pa.field("element", pa.date32(), nullable))),
        "t": pa.array(["row %d" % i for i in range(n)]),
    }, schema=pa.schema([pa.field("k", pa.int64(), False), pa.field("s", st_type, nullable), pa.field("li", pa.list_(pa.field("element", pa.int64(), nullable)), nullable),
                         pa.field("f", pa.int32(), nullable), pa.field("lf", pa.list_(pa.field("element", pa.float64(), nullable)), nullable),
                         pa.field("ld", pa.list_(pa.field("element", pa.date32(), nullable)), nullable), pa.field("t", pa.string(), False)]))


def _types(schema):
    return [S.from_arrow_type(f.type) for f in schema]


def _run(plan, ncols, conf=None):
    batches = native.execute_to_table([], ncols, plan.encode(), **({"config": S.config_map(conf)} if conf else {}))
    return pa.Table.from_batches(batches) if batches else None


def _same(got_col, want_col, what):
    g, w = got_col.combine_chunks().to_pylist(), want_col.combine_chunks().to_pylist()
    assert len(g) == len(w), what
    for i, (x, y) in enumerate(zip(g, w)):
        assert x == y, f"{what}: row {i}: {x!r} != {y!r}"


def _scan_and_compare(path, t):
    want = papq.read_table(path)
    got = _run(S.native_scan([path], t.schema.names, _types(t.schema)), t.num_columns)
    assert got.num_rows == want.num_rows
    for i, name in enumerate(t.schema.names):
        assert got.column(i).type == want.column(name).type or pa.types.is_nested(want.column(name).type), name
        _same(got.column(i), want.column(name), name)
    return got


@pytest.mark.parametrize("codec,version,dictionary", [("none", "1.0", True), ("snappy", "1.0", False), ("zstd", "2.0", True), ("snappy", "2.0", True), ("gzip", "1.0", True)])
def test_scan_of_struct_and_list_columns(built, tmp_path, codec, version, dictionary):
    """several row groups, small pages (lists cross them), NULL structs / fields / lists / elements, empty lists, next to flat columns"""
    t = _nested_table(20_000, 41)
    path = str(tmp_path / f"nested_{codec}_{version}.parquet")
    papq.write_table(t, path, compression=codec, data_page_version=version, use_dictionary=dictionary, row_group_size=6_000, data_page_size=8 << 10)
    got = _scan_and_compare(path, t)
    assert pa.types.is_struct(got.column(1).type) and pa.types.is_list(got.column(2).type)


def test_required_structs_fields_lists_and_elements(built, tmp_path):
    """nothing optional: no definition levels for the struct's fields, one level for the lists (an element slot or an empty list)"""
    t = _nested_table(5_000, 42, nullable=False)
    path = str(tmp_path / "nested_required.parquet")
    papq.write_table(t, path, compression="snappy", row_group_size=2_000, data_page_size=16 << 10)
    _scan_and_compare(path, t)


def test_a_subset_of_the_struct_fields_in_another_order(built, tmp_path):
    """the requested struct names fields c, a of the file's (a, b, c, d, e): Spark's clipped schema — matched by name"""
    t = _nested_table(3_000, 43)
    path = str(tmp_path / "nested_subset.parquet")
    papq.write_table(t, path, compression="zstd", row_group_size=1_000)
    st = S.struct_type([("c", S.T_DOUBLE, True), ("a", S.T_INT32, True)])
    got = _run(S.native_scan([path], ["s", "k"], [st, S.T_INT64]), 2)
    want = papq.read_table(path)
    ws = want.column("s").combine_chunks()
    exp = [None if v is None else {"c": v["c"], "a": v["a"]} for v in ws.to_pylist()]
    assert got.column(0).combine_chunks().to_pylist() == exp
    _same(got.column(1), want.column("k"), "k")


def test_get_struct_field_and_passthrough_in_a_projection(built, tmp_path):
    t = _nested_table(8_000, 44)
    path = str(tmp_path / "nested_proj.parquet")
    papq.write_table(t, path, compression="snappy", row_group_size=3_000, data_page_size=32 << 10)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    s = S.col(1, ty[1])
    plan = S.project(scan, [S.col(0, ty[0]), S.get_struct_field(s, 0), S.get_struct_field(s, 1), S.get_struct_field(s, 3), S.col(2, ty[2]), s, S.get_struct_field(s, 4)])
    got = _run(plan, 7)
    want = papq.read_table(path)
    ws = want.column("s").combine_chunks()
    _same(got.column(0), want.column("k"), "k")
    for out, fld in ((1, "a"), (2, "b"), (3, "d"), (6, "e")):
        exp = [None if v is None else v[fld] for v in ws.to_pylist()]
        assert got.column(out).combine_chunks().to_pylist() == exp, fld
    _same(got.column(4), want.column("li"), "li")
    _same(got.column(5), want.column("s"), "s")


def test_filter_passes_nested_columns_through(built, tmp_path):
    """Filter on a flat column and on a struct's field; the struct and the lists of the surviving rows are gathered, children and all"""
    t = _nested_table(12_000, 45)
    path = str(tmp_path / "nested_filter.parquet")
    papq.write_table(t, path, compression="zstd", row_group_size=5_000, data_page_size=16 << 10)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    s = S.col(1, ty[1])
    pred = S.and_(S.lt(S.col(3, ty[3]), S.lit(40, S.T_INT32)), S.gt(S.get_struct_field(s, 0), S.lit(-500, S.T_INT32)))
    plan = S.project(S.filter_(scan, pred), [S.col(0, ty[0]), s, S.col(2, ty[2]), S.col(5, ty[5]), S.col(6, ty[6])])
    got = _run(plan, 5)
    want = papq.read_table(path)
    keep = [i for i, (f, sv) in enumerate(zip(want.column("f").to_pylist(), want.column("s").to_pylist()))
            if f is not None and f < 40 and sv is not None and sv["a"] is not None and sv["a"] > -500]
    assert got.num_rows == len(keep) and len(keep) > 100
    wt = want.take(pa.array(keep, pa.int64()))
    for out, name in enumerate(["k", "s", "li", "ld", "t"]):
        _same(got.column(out), wt.column(name), name)


def test_what_the_nested_scan_refuses(built, tmp_path):
    t = pa.table({"m": pa.array([[("a", 1)], None], pa.map_(pa.string(), pa.int32())),
                  "ss": pa.array([{"x": {"y": 1}}, None], pa.struct([("x", pa.struct([("y", pa.int32())]))])),
                  "ls": pa.array([[[1]], None], pa.list_(pa.list_(pa.int32())))})
    path = str(tmp_path / "nested_refused.parquet")
    papq.write_table(t, path)
    deep = S.struct_type([("x", S.struct_type([("y", S.T_INT32, True)]), True)])
    for names, types, msg in ((["ss"], [deep], "deeper than one level"), (["ls"], [S.list_type(S.list_type(S.T_INT32))], "lists of List"),
                              (["ss"], [S.T_INT32], "is a group")):
        with pytest.raises((native.CometNativeException, native.CometQueryExecutionException), match=msg):
            _run(S.native_scan([path], names, types), len(names))


@pytest.mark.parametrize("codec", [S.CODEC_NONE, S.CODEC_ZSTD])
def test_shuffle_writer_carries_nested_columns(built, tmp_path, codec):
    """ShuffleWriter over a scan with struct and list columns, hash-partitioned on a flat key: rows are taken partition-major on the device —
    children and all —, nested columns come to the host whole, flat ones in slabs; every block decoded by pyarrow's IPC reader holds the rows
    the oracle's partitioner assigns, in order"""
    import struct as pystruct
    from oracle import shuffle_oracle as SO
    t = _nested_table(15_000, 46)
    path = str(tmp_path / "nested_shuffle.parquet")
    papq.write_table(t, path, compression="snappy", row_group_size=4_000, data_page_size=32 << 10)
    ty = _types(t.schema)
    data, index = str(tmp_path / "shuffle.data"), str(tmp_path / "shuffle.index")
    plan = S.shuffle_writer(S.native_scan([path], t.schema.names, ty), data, index, partitioning="hash", hash_exprs=[S.col(0, ty[0])], num_partitions=5, codec=codec)
    assert native.execute_to_table([], 0, plan.encode(), batch_size=2048) == []
    want = papq.read_table(path)
    flat = want.select(["k"])
    _, _, rows = SO.shuffle_write(S, flat, "hash", [0], 5, 2048)
    raw, idx = open(data, "rb").read(), open(index, "rb").read()
    offs = pystruct.unpack("<6q", idx)
    assert offs[0] == 0 and offs[-1] == len(raw)
    total = 0
    for p in range(5):
        blocks = SO.read_partition(raw, idx, p)
        exp = want.take(pa.array(rows[p]))
        n = exp.num_rows
        assert [b.num_rows for b in blocks] == [2048] * (n // 2048) + ([n % 2048] if n % 2048 else [])
        if blocks:
            got = pa.Table.from_batches(blocks)
            for c, name in enumerate(t.schema.names):
                _same(got.column(c), exp.column(name), f"partition {p} column {name}")
        total += n
    assert total == want.num_rows


def test_limit_and_sort_take_nested_rows(built, tmp_path):
    """Sort on a flat key / Limit above a scan with nested columns: the row gather (take_rows) moves struct and list rows with their children"""
    t = _nested_table(6_000, 47)
    path = str(tmp_path / "nested_sort.parquet")
    papq.write_table(t, path, compression="zstd", row_group_size=2_500)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    want = papq.read_table(path)
    got = _run(S.limit(scan, 1000, 37), t.num_columns)
    for c, name in enumerate(t.schema.names):
        _same(got.column(c), want.slice(37, 1000 - 37).column(name), name)
    # descending by k (unique): the reversed table
    plan = S.sort(scan, [(S.col(0, ty[0]), True, False)])
    got = _run(plan, t.num_columns)
    order = pa.array(np.arange(want.num_rows - 1, -1, -1, dtype=np.int64))
    rev = want.take(order)
    for c, name in enumerate(["k", "s", "li"]):
        _same(got.column(c), rev.column(name), name)


def test_nested_columns_of_a_host_stream_input(built):
    """Scan (what the JVM hands over as ArrowArrayStream batches) with struct and list fields — lists of strings and of structs, a struct
    inside a struct too: the batches of a chunk are concatenated on the host and uploaded with their children; Filter on a flat column and on a
    struct's field, GetStructField, passthrough"""
    from tests.test_shuffle_nested_cpu import _batch
    b = _batch(9_000, 61)
    t = pa.Table.from_batches([b])
    inner = pa.array([None if i % 7 == 0 else {"p": i, "q": {"r": "r%d" % i, "z": None if i % 3 == 0 else i * 0.5}} for i in range(t.num_rows)],
                     pa.struct([("p", pa.int64()), ("q", pa.struct([("r", pa.string()), ("z", pa.float64())]))]))
    t = t.append_column("deep", inner)
    ty = _types(t.schema)
    names = t.schema.names
    s = S.col(names.index("s"), ty[names.index("s")])
    scan = S.scan(ty)
    pred = S.and_(S.gt_eq(S.col(0, ty[0]), S.lit(100, S.T_INT64)), S.lt(S.get_struct_field(s, 0), S.lit(600, S.T_INT32)))
    outs = [S.col(i, ty[i]) for i in range(len(ty))] + [S.get_struct_field(s, 1), S.get_struct_field(S.col(names.index("deep"), ty[names.index("deep")]), 0)]
    plan = S.project(S.filter_(scan, pred), outs)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t, 1000)], len(outs), plan.encode()))
    keep = [i for i, (k, sv) in enumerate(zip(t.column("k").to_pylist(), t.column("s").to_pylist())) if k >= 100 and sv is not None and sv["a"] is not None and sv["a"] < 600]
    assert got.num_rows == len(keep) and len(keep) > 500
    want = t.take(pa.array(keep, pa.int64()))
    for c, name in enumerate(names):
        _same(got.column(c), want.column(name), name)
    assert got.column(len(names)).to_pylist() == [v["b"] for v in want.column("s").to_pylist()]
    assert got.column(len(names) + 1).to_pylist() == [None if v is None else v["p"] for v in want.column("deep").to_pylist()]


def test_nested_columns_round_trip_through_shuffle_files(built, tmp_path):
    """stage 1 writes shuffle files with struct / list columns (GPU writer), stage 2 reads every partition back through a ShuffleScan leaf and
    passes the rows on: the union of the partitions is the table"""
    t = _nested_table(10_000, 48)
    path = str(tmp_path / "nested_rt.parquet")
    papq.write_table(t, path, compression="zstd", row_group_size=3_000)
    ty = _types(t.schema)
    data, index = str(tmp_path / "shuffle.data"), str(tmp_path / "shuffle.index")
    P = 4
    plan = S.shuffle_writer(S.native_scan([path], t.schema.names, ty), data, index, partitioning="hash", hash_exprs=[S.col(0, ty[0])], num_partitions=P, codec=S.CODEC_LZ4)
    assert native.execute_to_table([], 0, plan.encode(), batch_size=1500) == []
    back = S.project(S.shuffle_scan(ty), [S.col(i, ty[i]) for i in range(len(ty))])
    rows = {}
    for p in range(P):
        out = native.execute_to_table([native.ShuffleBlockInput.from_files(data, index, p)], len(ty), back.encode(), batch_size=0)
        for b in out:
            cols = [c.to_pylist() for c in b.columns]
            for i in range(b.num_rows):
                rows[cols[0][i]] = tuple(cols[j][i] for j in range(1, len(cols)))
    want = papq.read_table(path)
    wcols = [want.column(n).to_pylist() for n in t.schema.names]
    assert len(rows) == want.num_rows
    for i in range(want.num_rows):
        assert rows[wcols[0][i]] == tuple(wcols[j][i] for j in range(1, len(wcols))), i


@pytest.mark.parametrize("codec,version", [("snappy", "1.0"), ("zstd", "2.0")])
def test_lists_of_strings_and_booleans(built, tmp_path, codec, version):
    """elements that are not one fixed-width value each: the element column is TAKEN out of the leaf's column over entries (offsets + bytes, or
    bits) by the entries that hold a slot — dictionary-encoded and PLAIN strings, long and empty ones, NULL elements, empty and NULL lists"""
    rng = np.random.default_rng(49)
    n = 12_000
    words = ["", "a", "bb", "a considerably longer string that does not fit any packed form", "日本語", "x" * 300]

    def lst(make, pnull=0.1, pel=0.15):
        out = []
        for _ in range(n):
            if rng.random() < pnull:
                out.append(None)
                continue
            k = int(rng.integers(0, 7)) if rng.random() > 0.03 else 200
            out.append([None if rng.random() < pel else make() for _ in range(k)])
        return out

    t = pa.table({
        "k": pa.array(np.arange(n, dtype=np.int64)),
        "ls": pa.array(lst(lambda: words[int(rng.integers(0, len(words)))]), pa.list_(pa.string())),                 # few distinct values: dictionary pages
        "lu": pa.array(lst(lambda: "u%d" % int(rng.integers(0, 10**9))), pa.list_(pa.string())),                      # unique values: falls back to PLAIN
        "lb": pa.array(lst(lambda: bool(rng.integers(0, 2))), pa.list_(pa.bool_())),
        "lr": pa.array(lst(lambda: "r%d" % int(rng.integers(0, 50)), pnull=0.0, pel=0.0), pa.list_(pa.field("element", pa.string(), False)), ),
    })
    path = str(tmp_path / f"lists_{codec}.parquet")
    papq.write_table(t, path, compression=codec, data_page_version=version, row_group_size=5_000, data_page_size=16 << 10)
    _scan_and_compare(path, t)
    ty = _types(t.schema)
    plan = S.project(S.filter_(S.native_scan([path], t.schema.names, ty), S.lt(S.col(0, ty[0]), S.lit(3_000, S.T_INT64))), [S.col(1, ty[1]), S.col(3, ty[3])])
    got = _run(plan, 2)
    want = papq.read_table(path).slice(0, 3_000)
    _same(got.column(0), want.column("ls"), "ls")
    _same(got.column(1), want.column("lb"), "lb")


def _explode_ref(rows_k, lists, outer, position):
    """Spark's explode / posexplode [_outer] restated on Python lists: (carried value, [pos,] element) per element; a NULL or empty list yields
    nothing — or one row with NULL position and element under *_outer (GenerateExec; the reference: planner.rs:1949-2110 over UnnestExec)"""
    out = []
    for k, l in zip(rows_k, lists):
        if not l:
            if outer:
                out.append((k, None, None) if position else (k, None))
            continue
        for j, e in enumerate(l):
            out.append((k, j, e) if position else (k, e))
    return out


@pytest.mark.parametrize("outer", [False, True])
@pytest.mark.parametrize("position", [False, True])
def test_explode_of_list_columns(built, tmp_path, outer, position):
    t = _nested_table(7_000, 50)
    path = str(tmp_path / "explode.parquet")
    papq.write_table(t, path, compression="snappy", row_group_size=2_500)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    want = papq.read_table(path)
    ks = want.column("k").to_pylist()
    for col, name in ((2, "li"), (5, "ld")):
        plan = S.explode(scan, S.col(col, ty[col]), [S.col(0, ty[0])], outer=outer, position=position)
        got = _run(plan, 3 if position else 2)
        exp = _explode_ref(ks, want.column(name).to_pylist(), outer, position)
        rows = list(zip(*[got.column(c).to_pylist() for c in range(got.num_columns)]))
        assert rows == exp, name
    # carried: a struct (gathered with its children), one of its fields, a string; Filter + Projection above the Explode
    s = S.col(1, ty[1])
    ex = S.explode(scan, S.col(4, ty[4]), [S.col(0, ty[0]), s, S.get_struct_field(s, 1), S.col(6, ty[6])], outer=outer, position=False)
    plan = S.project(S.filter_(ex, S.lt(S.col(0, S.T_INT64), S.lit(2_000, S.T_INT64))), [S.col(0, S.T_INT64), S.col(1, ty[1]), S.col(2, S.T_STRING), S.col(3, S.T_STRING), S.col(4, S.T_DOUBLE)])
    got = _run(plan, 5)
    exp = []
    for k, sv, tv, l in zip(ks, want.column("s").to_pylist(), want.column("t").to_pylist(), want.column("lf").to_pylist()):
        if k >= 2_000:
            continue
        for e in (l if l else ([None] if outer else [])):
            exp.append((k, sv, None if sv is None else sv["b"], tv, e))
    rows = list(zip(*[got.column(c).to_pylist() for c in range(5)]))
    assert rows == exp


def test_explode_of_lists_of_strings_and_structs(built):
    """element types that are gathered, not copied: strings (Parquet scan) and structs (a host-stream input)"""
    from tests.test_shuffle_nested_cpu import _batch
    b = _batch(4_000, 62)
    t = pa.Table.from_batches([b])
    ty = _types(t.schema)
    names = t.schema.names
    for name in ("ls", "lst"):
        c = names.index(name)
        plan = S.explode(S.scan(ty), S.col(c, ty[c]), [S.col(0, ty[0])], outer=True, position=True)
        got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t, 700)], 3, plan.encode()))
        exp = _explode_ref(t.column("k").to_pylist(), t.column(name).to_pylist(), True, True)
        rows = list(zip(*[got.column(i).to_pylist() for i in range(3)]))
        assert rows == exp, name


@pytest.mark.parametrize("codec,version,nullable", [("snappy", "1.0", True), ("zstd", "2.0", True), ("none", "1.0", False)])
def test_lists_of_structs_out_of_parquet(built, tmp_path, codec, version, nullable):
    """array<struct<…>>: the fields' leaves share one repeated group; offsets / list validity / the element struct's validity come from the first
    leaf's levels, every field is taken out of its leaf by the slots' entries.  NULL lists, empty lists, NULL element structs, NULL fields,
    strings and decimals among the fields, lists that cross pages; then Explode and a Filter above it"""
    rng = np.random.default_rng(63)
    n = 9_000
    ft = pa.struct([pa.field("x", pa.int64(), nullable), pa.field("y", pa.string(), nullable), pa.field("z", pa.decimal128(10, 2), nullable), pa.field("w", pa.bool_(), nullable)])
    lt = pa.list_(pa.field("element", ft, nullable))

    def maybe(p, f):
        return f() if (not nullable or rng.random() >= p) else None

    def elem():
        return {"x": maybe(0.1, lambda: int(rng.integers(-10**9, 10**9))), "y": maybe(0.15, lambda: "y%d" % int(rng.integers(0, 40)) if rng.random() < 0.8 else "a longer value of y %d" % int(rng.integers(0, 10**6))),
                "z": maybe(0.1, lambda: Decimal(int(rng.integers(-10**7, 10**7))).scaleb(-2)), "w": maybe(0.1, lambda: bool(rng.integers(0, 2)))}

    lists = []
    for _ in range(n):
        k = int(rng.integers(0, 6)) if rng.random() > 0.02 else 150
        lists.append(maybe(0.1, lambda: [maybe(0.12, elem) for _ in range(k)]))
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "ls": pa.array(lists, lt), "v": pa.array(rng.integers(0, 100, n).astype(np.int32))},
                 schema=pa.schema([pa.field("k", pa.int64(), False), pa.field("ls", lt, nullable), pa.field("v", pa.int32(), False)]))
    path = str(tmp_path / f"list_struct_{codec}.parquet")
    papq.write_table(t, path, compression=codec, data_page_version=version, row_group_size=4_000, data_page_size=16 << 10)
    _scan_and_compare(path, t)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    want = papq.read_table(path)
    ex = S.explode(S.filter_(scan, S.lt(S.col(2, ty[2]), S.lit(50, S.T_INT32))), S.col(1, ty[1]), [S.col(0, ty[0])], outer=True, position=True)
    got = _run(ex, 3)
    keep = [i for i, v in enumerate(want.column("v").to_pylist()) if v < 50]
    exp = _explode_ref([want.column("k")[i].as_py() for i in keep], [want.column("ls")[i].as_py() for i in keep], True, True)
    rows = list(zip(*[got.column(c).to_pylist() for c in range(3)]))
    assert rows == exp


@pytest.mark.parametrize("codec,version", [("snappy", "1.0"), ("zstd", "2.0")])
def test_map_columns_out_of_parquet_and_through_a_shuffle(built, tmp_path, codec, version):
    """map<K, V>: in the file a MAP group of one repeated key_value group (required key, optional value) — read like a list of (key, value)
    structs, laid out in HBM like Arrow's Map (offsets + entries); exported as `+m`, passed through a Filter, written to and read back from
    shuffle files"""
    rng = np.random.default_rng(64)
    n = 8_000

    def mk(make_v, pnull=0.1):
        out = []
        for _ in range(n):
            if rng.random() < pnull:
                out.append(None)
                continue
            k = int(rng.integers(0, 6)) if rng.random() > 0.02 else 120
            out.append([("key-%d" % j, None if rng.random() < 0.15 else make_v()) for j in range(k)])
        return out

    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)),
                  "m": pa.array(mk(lambda: int(rng.integers(-10**9, 10**9))), pa.map_(pa.string(), pa.int64())),
                  "ms": pa.array(mk(lambda: "v%d" % int(rng.integers(0, 30))), pa.map_(pa.string(), pa.string())),
                  "mi": pa.array([None if rng.random() < 0.1 else [(int(j), float(j) * 0.5) for j in range(int(rng.integers(0, 4)))] for _ in range(n)], pa.map_(pa.int32(), pa.float64()))})
    path = str(tmp_path / f"maps_{codec}.parquet")
    papq.write_table(t, path, compression=codec, data_page_version=version, row_group_size=3_000, data_page_size=16 << 10)
    got = _scan_and_compare(path, t)
    assert pa.types.is_map(got.column(1).type)
    ty = _types(t.schema)
    scan = S.native_scan([path], t.schema.names, ty)
    want = papq.read_table(path)
    plan = S.project(S.filter_(scan, S.gt_eq(S.col(0, ty[0]), S.lit(5_000, S.T_INT64))), [S.col(1, ty[1]), S.col(3, ty[3]), S.col(0, ty[0])])
    got = _run(plan, 3)
    _same(got.column(0), want.slice(5_000).column("m"), "m")
    _same(got.column(1), want.slice(5_000).column("mi"), "mi")
    data, index = str(tmp_path / "shuffle.data"), str(tmp_path / "shuffle.index")
    assert native.execute_to_table([], 0, S.shuffle_writer(scan, data, index, partitioning="hash", hash_exprs=[S.col(0, ty[0])], num_partitions=3, codec=S.CODEC_ZSTD).encode(), batch_size=2000) == []
    back = S.project(S.shuffle_scan(ty), [S.col(i, ty[i]) for i in range(len(ty))])
    rows = {}
    for p in range(3):
        for b in native.execute_to_table([native.ShuffleBlockInput.from_files(data, index, p)], len(ty), back.encode(), batch_size=0):
            cols = [c.to_pylist() for c in b.columns]
            for i in range(b.num_rows):
                rows[cols[0][i]] = tuple(cols[j][i] for j in range(1, len(cols)))
    wcols = [want.column(nm).to_pylist() for nm in t.schema.names]
    assert len(rows) == n
    for i in range(n):
        assert rows[wcols[0][i]] == tuple(wcols[j][i] for j in range(1, len(wcols))), i
