"""Host-side logic of native.DeviceTable that needs no GPU (the tensors live on the CPU here): the fixed-length declaration of Utf8 columns
(comet:utf8_fixed_len, read by exec_input.cpp validate_input_schema) is made exactly for the columns whose values all have one length ≤ 15,
and it reaches the ArrowSchema a DeviceInput hands to the library."""
import ctypes

import numpy as np
import pyarrow as pa

from datafusion_comet_amd import native

KEY = b"comet:utf8_fixed_len"


def _hints(table):
    dt = native.DeviceTable.from_arrow(table, device="cpu").with_string_hints()
    return {f.name: (f.metadata or {}).get(KEY) for f in dt.schema}, dt


def test_fixed_length_columns_are_declared_and_others_are_not():
    n = 5000
    rng = np.random.default_rng(3)
    t = pa.table({
        "flag": pa.array([["A", "N", "R"][i] for i in rng.integers(0, 3, n)]),                     # CHAR(1)
        "code": pa.array(["%04d" % i for i in rng.integers(0, 10000, n)]),                          # always 4 bytes
        "name": pa.array(["x" * int(i) for i in rng.integers(0, 9, n)]),                            # lengths vary
        "mostly": pa.array(["ab"] * (n - 1) + ["abc"]),                                             # one odd value at the very end
        "swap": pa.array(["abc", "a"] + ["ab"] * (n - 2)),                                          # the total fits 2·n, the values do not
        "long": pa.array(["0123456789abcdefXYZ"] * n),                                              # uniform but longer than 15 bytes
        "nulls": pa.array(["Z" if i % 7 else None for i in range(n)]),                              # NULL rows occupy 0 bytes: not uniform
        "num": pa.array(rng.integers(0, 100, n)),
        "empty": pa.array([""] * n),
    })
    got, dt = _hints(t)
    assert got == {"flag": b"1", "code": b"4", "name": None, "mostly": None, "swap": None, "long": None, "nulls": None, "num": None, "empty": b"0"}
    # the data is shared, not copied, and other metadata survives
    base = native.DeviceTable.from_arrow(t.replace_schema_metadata({b"k": b"v"}), device="cpu")
    hinted = base.with_string_hints()
    assert hinted.values[0].data_ptr() == base.values[0].data_ptr() and hinted.schema.metadata == {b"k": b"v"}


def test_no_rows_no_declaration():
    got, _ = _hints(pa.table({"s": pa.array([], pa.string())}))
    assert got == {"s": None}


def test_the_declaration_reaches_the_exported_arrow_schema():
    _, dt = _hints(pa.table({"flag": pa.array(["A", "B", "C"]), "v": pa.array([1, 2, 3])}))
    inp = native.DeviceInput(dt)
    # through the C Data Interface, the way the library reads it: the exported struct is re-imported by pyarrow
    c_schema = native.ArrowSchemaC()
    ptr = ctypes.addressof(c_schema)
    assert inp._get_schema(None, ptr) == 0
    # the raw metadata block of child 0 is what exec_input.cpp parses: int32 pair count, then (int32 length, bytes) for key and value
    child = c_schema.children[0].contents
    md_ptr = ctypes.c_void_p.from_address(ctypes.addressof(child) + native.ArrowSchemaC.metadata.offset).value    # the field is a char*: read the address, not a str
    raw = ctypes.string_at(md_ptr, 4 + 4 + len(KEY) + 4 + 1)
    assert raw == (1).to_bytes(4, "little") + len(KEY).to_bytes(4, "little") + KEY + (1).to_bytes(4, "little") + b"1"
    back = pa.Schema._import_from_c(ptr)
    assert back.field("flag").metadata == {KEY: b"1"} and not back.field("v").metadata
