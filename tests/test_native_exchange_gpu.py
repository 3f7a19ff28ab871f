"""The exchange inside libcomet.so (csrc/exchange.cpp): several ranks meet through the in-process transport (threads of one
process — the Spark-executor shape; here all on the one GPU of the test box), every rank receives exactly the rows Spark's HashPartitioning
sends it (murmur3 seed 42 → pmod, the oracle's restatement), sender after sender, each sender's rows in input order; NULL keys hash as Spark
hashes them; validity survives; Utf8 columns arrive with rebuilt offsets, Boolean columns re-packed.  The RCCL transport is driven with a 1-rank communicator (AllGather + a send/recv group to itself): symbol
binding, datatype constants and group semantics on real hardware; the N-rank run is bench.py --gpus N (tools/q3_dist.py)."""
import threading

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu


def _shard(seed, n):
    rng = np.random.default_rng(seed)
    return pa.table({
        "k": pa.array(rng.integers(-10**12, 10**12, n), pa.int64(), mask=rng.random(n) < 0.05),
        "d": pa.array(rng.integers(8000, 12000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "v": pa.array([__import__("decimal").Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**10, 10**10, n)], pa.decimal128(12, 2),
                      mask=(rng.random(n) < 0.1) if seed % 2 else None),
        "f": pa.array(rng.standard_normal(n)),
        # Utf8 (empty values, multi-byte characters, a few long ones) and Boolean travel too: lengths + bytes, one byte per bit
        "s": pa.array([("" if x % 11 == 0 else "né" * (x % 5) + str(x) + ("/" + "z" * 300 if x % 97 == 0 else "")) for x in rng.integers(0, 5000, n)],
                      pa.string(), mask=rng.random(n) < 0.07),
        "b": pa.array(rng.random(n) < 0.5, pa.bool_(), mask=(rng.random(n) < 0.2) if seed % 3 == 0 else None),
    })


def _expected(shards, world, key_cols):
    from oracle import oracle as O
    out = []
    for r in range(world):
        parts = []
        for sh in shards:
            pids = O.hash_partition_ids(S, sh, key_cols, world)
            parts.append(sh.filter(pa.array(pids == r)))
        out.append(pa.concat_tables(parts))
    return out


@pytest.mark.parametrize("world,keys", [(2, [0]), (4, [0, 1]), (3, [2]), (3, [4]), (4, [5, 4, 0])])
def test_local_transport_delivers_sparks_partitions(built, world, keys):
    shards = [_shard(100 + r, 20_000 + 777 * r) for r in range(world)]
    shards[-1] = shards[-1].slice(0, 0) if world == 3 else shards[-1]           # an empty sender
    want = _expected(shards, world, keys)
    got, errs = [None] * world, []
    group = 7000 + world * 16 + sum(keys)

    def rank_main(r):
        try:
            comm = native.NativeComm(world, r, 0, local_group=group)
            dt = native.DeviceTable.from_arrow(shards[r])
            got[r] = comm.exchange(dt, keys).to_arrow()
            again = comm.exchange(dt, keys).to_arrow()                        # a communicator is reusable
            assert again.equals(got[r])
            comm.close()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    for r in range(world):
        assert got[r].num_rows == want[r].num_rows
        for c in range(6):
            assert got[r].column(c).to_pylist() == want[r].column(c).to_pylist(), (r, c)


def test_rccl_transport_with_a_one_rank_communicator(built, monkeypatch):
    monkeypatch.setenv("COMET_EXCHANGE_FORCE_RCCL", "1")
    sh = _shard(5, 30_000)
    comm = native.NativeComm(1, 0, 0, unique_id=native.NativeComm.unique_id())
    out = comm.exchange(native.DeviceTable.from_arrow(sh), [0]).to_arrow()
    comm.close()
    for c in range(6):
        assert out.column(c).to_pylist() == sh.column(c).to_pylist()


def test_tcp_transport_between_processes_on_the_gpu(built, tmp_path):
    """comet_comm_init_tcp: two and three PROCESSES (own HIP contexts, the one GPU of the test box) exchange through sockets — HBM
    buffers staged through pinned memory, the orchestration of exchange_core.hpp unchanged — and every rank receives Spark's partition."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import pyarrow as pa
from datafusion_comet_amd import native
import test_native_exchange_gpu as T
world, rank, peers, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
native.lib()
sh = T._shard(300 + rank, 9000 + 500 * rank)
comm = native.NativeComm(world, rank, 0, tcp_peers=peers, timeout_ms=30000)
assert comm.transport == "tcp"
got = comm.exchange(native.DeviceTable.from_arrow(sh), [0, 4]).to_arrow()
comm.close()
with pa.OSFile(out, "wb") as f, pa.ipc.new_file(f, got.schema) as w:
    w.write_table(got)
""" % (root, os.path.join(root, "tests"))
    for world in (2, 3):
        socks = [socket.socket() for _ in range(world)]
        for s_ in socks:
            s_.bind(("127.0.0.1", 0))
        peers = ",".join("127.0.0.1:%d" % s_.getsockname()[1] for s_ in socks)
        for s_ in socks:
            s_.close()
        procs = [subprocess.Popen([sys.executable, "-c", child, str(world), str(r), peers, str(tmp_path / f"w{world}r{r}.arrow")], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = [p.communicate(timeout=300)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), [l[-600:] for l in logs]
        want = _expected([_shard(300 + r, 9000 + 500 * r) for r in range(world)], world, [0, 4])
        for r in range(world):
            got = pa.ipc.open_file(str(tmp_path / f"w{world}r{r}.arrow")).read_all()
            assert got.num_rows == want[r].num_rows
            for c in range(6):
                assert got.column(c).to_pylist() == want[r].column(c).to_pylist(), (world, r, c)
