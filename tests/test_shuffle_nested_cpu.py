"""Nested columns in Comet shuffle blocks (the Arrow IPC stream inside a block; native/shuffle/src/writers/shuffle_block_writer.rs:179-238,
ipc.rs:23-52 — the reference writes whatever arrow-ipc writes): struct and list columns through comet_encode_shuffle_block /
comet_decode_shuffle_block, the hand-written flatbuffers writer and reader refereed by pyarrow's IPC implementation in both directions."""
import io

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native


def _cols(b):
    return [c.to_pylist() for c in b.columns]


def _batch(n, seed):
    from tests.test_parquet_nested_gpu import _nested_table
    t = _nested_table(n, seed)
    rng = np.random.default_rng(seed)
    ls = [None if rng.random() < 0.1 else [None if rng.random() < 0.1 else "v%d" % int(x) for x in rng.integers(0, 99, int(rng.integers(0, 5)))] for _ in range(n)]
    lst = [None if rng.random() < 0.1 else [{"x": int(x), "y": None if x % 3 == 0 else "y%d" % int(x)} for x in rng.integers(0, 99, int(rng.integers(0, 4)))] for _ in range(n)]
    t = t.append_column("ls", pa.array(ls, pa.list_(pa.string())))
    t = t.append_column("lst", pa.array(lst, pa.list_(pa.struct([("x", pa.int64()), ("y", pa.string())]))))
    mp = [None if rng.random() < 0.1 else [("k%d" % j, None if rng.random() < 0.2 else float(rng.integers(0, 999)) / 4) for j in range(int(rng.integers(0, 5)))] for _ in range(n)]
    mnest = [None if rng.random() < 0.1 else [(int(j) * 3, [int(x) for x in rng.integers(0, 9, int(rng.integers(0, 3)))]) for j in range(int(rng.integers(0, 4)))] for _ in range(n)]
    t = t.append_column("mp", pa.array(mp, pa.map_(pa.string(), pa.float64())))
    t = t.append_column("mnest", pa.array(mnest, pa.map_(pa.int32(), pa.list_(pa.int64()))))
    return t.combine_chunks().to_batches()[0]


@pytest.mark.parametrize("codec", [0, 1, 2, 3])
def test_round_trip_of_nested_columns(built, codec):
    b = _batch(3_000, 51)
    blk = native.encode_shuffle_block(b, codec)
    got = native.decode_shuffle_block(blk[16:], b.num_columns)
    assert got.num_rows == b.num_rows
    assert _cols(got) == _cols(b)


def test_pyarrow_reads_what_the_writer_wrote_and_the_reader_reads_what_pyarrow_wrote(built):
    b = _batch(2_000, 52)
    blk = native.encode_shuffle_block(b, 0)
    assert blk[16:20] == b"NONE"
    rd = pa.ipc.open_stream(blk[20:]).read_all()
    assert rd.num_rows == b.num_rows and _cols(rd.combine_chunks().to_batches()[0]) == _cols(b)
    for f, want in zip(rd.schema, b.schema):      # the types, children included (field names are the writer's own: c0, c1, …)
        assert f.type == want.type or pa.types.is_nested(want.type)
    assert pa.types.is_struct(rd.schema[1].type) and [k.name for k in rd.schema[1].type] == ["a", "b", "c", "d", "e"]
    sink = io.BytesIO()
    w = pa.ipc.new_stream(sink, b.schema)
    w.write_batch(b)
    w.close()
    got = native.decode_shuffle_block(b"NONE" + sink.getvalue(), b.num_columns)
    assert _cols(got) == _cols(b)


def test_slices_of_nested_columns(built):
    """a block of rows [first, first + rows) of a longer batch: list offsets are rebased, the elements are the addressed ones only"""
    b = _batch(1_000, 53)
    for first, rows in ((0, 1), (7, 129), (500, 500), (999, 1)):
        s = b.slice(first, rows)
        got = native.decode_shuffle_block(native.encode_shuffle_block(s, 2)[16:], b.num_columns)
        assert _cols(got) == _cols(s), (first, rows)
    sink = io.BytesIO()
    s = b.slice(13, 200)
    w = pa.ipc.new_stream(sink, s.schema)
    w.write_batch(s)      # pyarrow writes a slice's list offsets as they are (not starting at 0) or rebased, depending on the version: both read
    w.close()
    got = native.decode_shuffle_block(b"NONE" + sink.getvalue(), b.num_columns)
    assert _cols(got) == _cols(s)


@pytest.mark.parametrize("codec", [0, 2])
def test_corrupted_nested_blocks_fail_cleanly(built, codec):
    """blocks with struct / list columns, overwritten, truncated or extended: an exception or a structurally sound batch — never a crash or a read
    outside the block (child counts, list offsets against the element column, nesting depth are all checked)"""
    import random
    b = _batch(200, 54)
    blk = native.encode_shuffle_block(b, codec)[16:]
    rng = random.Random(2000 + codec)
    outcomes = {"ok": 0, "error": 0}
    for trial in range(600):
        bad = bytearray(blk)
        k = trial % 4
        if k == 0:
            bad = bad[: rng.randrange(0, len(bad))]
        elif k == 1:
            bad += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        else:
            lim = len(bad) if k == 2 else min(len(bad), 2500)          # everywhere / concentrated on the frame header and the flatbuffers (schema with children, nodes, buffers)
            for _ in range(rng.randrange(1, 4)):
                bad[rng.randrange(0, lim)] = rng.choice([0, 0xFF, 0x80, rng.randrange(256)])
        try:
            native.decode_shuffle_block(bytes(bad), b.num_columns).validate(full=True)
            outcomes["ok"] += 1
        except (native.CometNativeException, pa.ArrowInvalid):
            outcomes["error"] += 1
    assert outcomes["error"] > 0, outcomes


def test_nested_input_batches_are_concatenated_like_pyarrow_does(built):
    """the host step of a Scan / ShuffleScan leaf over nested columns (comet_concat_nested_column = exec_util.cpp append_nested_rows): batches and
    SLICES of batches of struct / list columns become one column equal to pyarrow's own concatenation; a field under a NULL struct comes out
    NULL whatever the producer left in its slot (pyarrow leaves a valid zero)"""
    b = _batch(2_500, 55)
    for name in ("s", "li", "ld", "ls", "lst", "mp", "mnest"):
        col = b.column(b.schema.get_field_index(name))
        parts = [col.slice(0, 700), col.slice(700, 1), col.slice(701, 0), col.slice(701, 1299), col.slice(2000, 500)]
        got = native.concat_nested_column(parts)
        got.validate(full=True)
        assert got.to_pylist() == col.to_pylist(), name
    s = b.column(b.schema.get_field_index("s"))
    got = native.concat_nested_column([s.slice(3, 1000), s.slice(1003, 500)])
    want = s.slice(3, 1500)
    nulls = [i for i, v in enumerate(want.to_pylist()) if v is None]
    assert len(nulls) > 50
    for f in range(got.type.num_fields):
        kid = got.field(f)
        assert all(not kid[i].is_valid for i in nulls), got.type.field(f).name      # masked by the struct's validity
    with pytest.raises(native.CometNativeException, match="not a nested type"):
        native.concat_nested_column([pa.array([1, 2, 3])])
