"""One rank of the CPU-only multi-process exchange test (tests/test_native_exchange_procs_cpu.py): TEST INFRASTRUCTURE.

Every rank builds the same table from the seed, keeps its row range, asks the ORACLE for the partition id of each of its rows
(oracle.hash_partition_ids: the pinned murmur3 + pmod restatement) and runs csrc/exchange_core.hpp's exchange over csrc/exchange_tcp.hpp's
transport through the host stand-in library (tests/exchange_host/host_exchange.cpp).  What it received goes to --out as an Arrow IPC file;
errors go to --out + ".err" (and exit code 3) so that the parent can assert on them."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import pyarrow as pa        # noqa: E402


class Col(ctypes.Structure):      # CometExchangeColumn (include/comet_amd.h)
    _fields_ = [("type_id", ctypes.c_int32), ("precision", ctypes.c_int32), ("values", ctypes.c_void_p), ("validity", ctypes.c_void_p), ("aux", ctypes.c_void_p)]


def full_table(seed: int, n: int) -> pa.Table:
    import decimal
    rng = np.random.default_rng(seed)
    m = lambda frac: rng.random(n) < frac
    words = ["", "a", "xy", "exchange", "a-considerably-longer-string-value-that-crosses-sixteen-bytes", "ß∂ƒ", "tail" * 20]
    only_first = np.zeros(n, bool)
    only_first[: max(1, n // 10)] = rng.random(max(1, n // 10)) < 0.5           # NULLs on the first ranks' rows only
    return pa.table({
        "k64": pa.array(rng.integers(-50, 50, n), pa.int64(), mask=m(0.1)),
        "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32), mask=only_first),
        "dec": pa.array([None if x else decimal.Decimal(int(v)).scaleb(-2) for x, v in zip(m(0.05), rng.integers(-10**11, 10**11, n))], pa.decimal128(12, 2)),
        "flag": pa.array(rng.random(n) < 0.5, pa.bool_(), mask=m(0.2)),
        "s": pa.array([None if x else words[int(i)] + str(int(j)) for x, i, j in zip(m(0.15), rng.integers(0, len(words), n), rng.integers(0, 30, n))], pa.string()),
        "d": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "f": pa.array(rng.random(n)),
        "b8": pa.array(rng.integers(-128, 127, n).astype(np.int8)),
        "bin": pa.array([bytes(rng.integers(0, 256, int(l), dtype=np.uint8).tolist()) for l in rng.integers(0, 9, n)], pa.binary()),
    })


def shard_bounds(n: int, world: int, layout: str):
    """row range of every rank; 'hole' leaves rank 1 without rows"""
    if layout == "hole" and world >= 3:
        cuts = [0, n // 3, n // 3] + [n // 3 + (n - n // 3) * (r - 1) // (world - 2) for r in range(2, world)] + [n]
        cuts = cuts[:world] + [n]
    else:
        cuts = [n * r // world for r in range(world)] + [n]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def normalized(t: pa.Table) -> pa.Table:
    """offset-0 buffers (an IPC round trip), one chunk per column"""
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t.combine_chunks())
    return pa.ipc.open_stream(sink.getvalue()).read_all().combine_chunks()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--ports", required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rows", type=int, default=5000)
    ap.add_argument("--layout", default="even")
    ap.add_argument("--keys", default="0,4")
    ap.add_argument("--out", required=True)
    ap.add_argument("--mode", default="normal", choices=["normal", "die_before_exchange", "silent", "wrong_collective"])
    ap.add_argument("--timeout-ms", type=int, default=20000)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--wire", default="tcp", choices=["tcp", "rccl"], help="rccl: the product's RcclTransportT over whatever COMET_RCCL_LIBRARY names (tests/fake_rccl/)")
    ap.add_argument("--id-file", default="", help="rccl: where rank 0 leaves the 128-byte unique id for the other ranks (the out-of-band control plane)")
    a = ap.parse_args()
    from datafusion_comet_amd import serde as S
    from oracle import oracle as O
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "exchange_host", "_build", "libcomet_exchange_host.so"))
    lib.xh_last_error.restype = ctypes.c_char_p
    lib.xh_comm_init_tcp.restype = ctypes.c_int64
    lib.xh_comm_init_tcp.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.xh_exchange.restype = ctypes.c_int64
    lib.xh_exchange.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.xh_result_rows.restype = ctypes.c_int64
    lib.xh_result_rows.argtypes = [ctypes.c_int64]
    lib.xh_result_column.argtypes = [ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4
    lib.xh_result_release.argtypes = [ctypes.c_int64]
    lib.xh_comm_destroy.argtypes = [ctypes.c_int64]

    def fail(msg):
        with open(a.out + ".err", "w") as f:
            f.write(msg)
        sys.exit(3)

    if a.wire == "rccl":
        lib.xh_comm_init_rccl.restype = ctypes.c_int64
        lib.xh_comm_init_rccl.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]
        lib.xh_comm_unique_id.argtypes = [ctypes.c_char_p]
        lib.xh_comm_stats.argtypes = [ctypes.c_int64, ctypes.c_void_p]
        lib.xh_comm_wire_counters.argtypes = [ctypes.c_int64, ctypes.c_void_p]
        idbuf = ctypes.create_string_buffer(128)
        if a.rank == 0:
            if lib.xh_comm_unique_id(idbuf) != 0:
                fail("unique id: " + lib.xh_last_error().decode())
            with open(a.id_file + ".tmp", "wb") as f:
                f.write(idbuf.raw)
            os.replace(a.id_file + ".tmp", a.id_file)
        else:
            t0 = time.time()
            while not os.path.exists(a.id_file):
                if time.time() - t0 > a.timeout_ms / 1000.0:
                    fail("init: no unique id from rank 0")
                time.sleep(0.02)
            idbuf.raw = open(a.id_file, "rb").read()
        comm = lib.xh_comm_init_rccl(idbuf.raw, a.world, a.rank)
    else:
        peers = ",".join("127.0.0.1:" + p for p in a.ports.split(","))
        comm = lib.xh_comm_init_tcp(peers.encode(), a.world, a.rank, a.timeout_ms)
    if not comm:
        fail("init: " + lib.xh_last_error().decode())
    if a.mode == "die_before_exchange":
        os._exit(0)                          # the process ends with its sockets open: the peers see EOF mid-exchange
    if a.mode == "silent":
        time.sleep(a.timeout_ms / 1000.0 * 3 + 2)
        os._exit(0)
    keys = [int(k) for k in a.keys.split(",")]
    outs = []
    for rnd in range(a.rounds):
        t = full_table(a.seed + rnd, a.rows)
        lo, hi = shard_bounds(t.num_rows, a.world, a.layout)[a.rank]
        shard = normalized(t.slice(lo, hi - lo))
        n = shard.num_rows
        pids = np.ascontiguousarray(O.hash_partition_ids(S, shard, keys, a.world)[:n].astype(np.int32)) if n else np.zeros(1, np.int32)
        ncols = shard.num_columns if a.mode != "wrong_collective" or a.rank != 1 else shard.num_columns - 1
        cols = (Col * ncols)()
        keep = []
        for i in range(ncols):
            arr = shard.column(i).chunk(0) if n else pa.array([], shard.schema.field(i).type)
            st = S.from_arrow_type(arr.type)
            bufs = arr.buffers()
            cols[i].type_id, cols[i].precision = st.type_id, st.precision
            cols[i].validity = bufs[0].address if (bufs[0] is not None and arr.null_count) else None
            cols[i].values = bufs[1].address if len(bufs) > 1 and bufs[1] is not None else None
            cols[i].aux = bufs[2].address if len(bufs) > 2 and bufs[2] is not None else None
            keep.append(bufs)
        h = lib.xh_exchange(comm, ncols, ctypes.cast(cols, ctypes.c_void_p), n, pids.ctypes.data)
        if not h:
            fail("exchange: " + lib.xh_last_error().decode())
        rows = lib.xh_result_rows(h)
        arrays = []
        for i in range(ncols):
            f = shard.schema.field(i)
            pv, pb, pa_, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
            lib.xh_result_column(h, i, ctypes.byref(pv), ctypes.byref(pb), ctypes.byref(pa_), ctypes.byref(nb))
            grab = lambda p, k: pa.py_buffer(ctypes.string_at(p.value, k)) if (p.value and k) else pa.py_buffer(b"")
            valid = grab(pb, (rows + 7) // 8) if pb.value else None
            if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
                arrays.append(pa.Array.from_buffers(f.type, rows, [valid, grab(pv, (rows + 1) * 4), grab(pa_, nb.value)]))
            elif pa.types.is_boolean(f.type):
                arrays.append(pa.Array.from_buffers(f.type, rows, [valid, grab(pv, (rows + 7) // 8)]))
            else:
                w = 16 if pa.types.is_decimal(f.type) else f.type.bit_width // 8
                arrays.append(pa.Array.from_buffers(f.type, rows, [valid, grab(pv, rows * w)]))
        lib.xh_result_release(h)
        outs.append(pa.Table.from_arrays(arrays, schema=pa.schema([shard.schema.field(i) for i in range(ncols)])))
    if a.wire == "rccl":
        import json
        st, wc = (ctypes.c_int64 * 4)(), (ctypes.c_int64 * 7)()
        lib.xh_comm_stats(comm, st)
        nw = lib.xh_comm_wire_counters(comm, wc)
        with open(a.out + ".stats", "w") as f:
            json.dump({"comm_count": st[0], "comm_rank": st[1], "bytes_sent": st[2], "bytes_received": st[3],
                       "wire": dict(zip(["allgathers", "groups", "sends", "recvs", "self_pairs", "bytes_out", "bytes_in"], list(wc))) if nw == 7 else None}, f)
    lib.xh_comm_destroy(comm)
    with pa.OSFile(a.out, "wb") as f, pa.ipc.new_file(f, outs[0].schema) as w:
        for o in outs:
            w.write_table(o if o.num_rows else o.schema.empty_table())
            if o.num_rows == 0:
                w.write_batch(pa.RecordBatch.from_pylist([], schema=o.schema))
    with open(a.out + ".rows", "w") as f:
        f.write(",".join(str(o.num_rows) for o in outs))


if __name__ == "__main__":
    main()
