"""The in-library hash exchange ACROSS PROCESSES on a CPU-only box (VERDICT r2 missing-1 / next-5).

csrc/exchange_core.hpp — the orchestration the RCCL path runs between GPUs: count exchange, splits, order of the collectives, validity
on any rank, the Utf8 byte split, the offset rebuild — is a template over the memory space and the wire.  Here 2 and 4 PROCESSES run it
over the product's TCP transport (csrc/exchange_tcp.hpp) with the host stand-in Ops of tests/exchange_host/ (plain loops instead of the
HIP kernels: test infrastructure, not in libcomet.so), and what every rank receives is compared with the oracle's Spark partition ids
(oracle.hash_partition_ids): rank r must hold, sender after sender in rank order, each sender's rows of partition r in their input order.
Negative tests: a peer whose process ends mid-exchange, a peer that goes silent, a peer in a different collective → an error naming
the peer, never a hang."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "tests", "exchange_host")
sys.path.insert(0, HOST_DIR)


@pytest.fixture(scope="module")
def host_lib():
    out = os.path.join(HOST_DIR, "_build")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", os.path.join(HOST_DIR, "host_exchange.cpp"), "-o",
                           os.path.join(out, "libcomet_exchange_host.so")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return True


def free_ports(n):
    socks = [socket.socket() for _ in range(n)]
    for s in socks:
        s.bind(("127.0.0.1", 0))
    ports = [s.getsockname()[1] for s in socks]
    for s in socks:
        s.close()
    return ports


def launch(world, tmp_path, modes=None, **kw):
    ports = ",".join(str(p) for p in free_ports(world))
    procs = []
    for r in range(world):
        cmd = [sys.executable, os.path.join(HOST_DIR, "rank_main.py"), "--world", str(world), "--rank", str(r), "--ports", ports, "--out", str(tmp_path / f"r{r}.arrow")]
        for k, v in kw.items():
            cmd += ["--" + k.replace("_", "-"), str(v)]
        if modes and modes.get(r):
            cmd += ["--mode", modes[r]]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    return procs


def wait_all(procs, timeout=120):
    t0 = time.time()
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung")
        logs.append(out)
    return logs


def expected(world, seed, rows, layout, keys, rank):
    import rank_main as RM
    from datafusion_comet_amd import serde as S
    from oracle import oracle as O
    t = RM.full_table(seed, rows)
    parts = []
    for lo, hi in RM.shard_bounds(rows, world, layout):
        shard = RM.normalized(t.slice(lo, hi - lo))
        if shard.num_rows == 0:
            continue
        pids = O.hash_partition_ids(S, shard, keys, world)[: shard.num_rows]
        parts.append(shard.filter(pa.array(pids == rank)))
    return pa.concat_tables(parts).combine_chunks() if parts else t.schema.empty_table()


@pytest.mark.parametrize("world,layout,rows,keys", [(2, "even", 5000, "0,4"), (4, "even", 6001, "0,4"), (4, "hole", 3000, "4"), (3, "even", 2, "0"), (4, "even", 40000, "2,3,5")])
def test_exchange_between_processes_matches_the_oracle_partitioning(host_lib, tmp_path, world, layout, rows, keys):
    procs = launch(world, tmp_path, seed=11, rows=rows, layout=layout, keys=keys, rounds=2)
    logs = wait_all(procs)
    for r, p in enumerate(procs):
        err = tmp_path / f"r{r}.arrow.err"
        assert p.returncode == 0, (logs[r][-800:], err.read_text() if err.exists() else "")
    total = 0
    for r in range(world):
        with pa.ipc.open_file(str(tmp_path / f"r{r}.arrow")) as f:
            batches = [f.get_batch(i) for i in range(f.num_record_batches)]
        counts = [int(x) for x in (tmp_path / f"r{r}.arrow.rows").read_text().split(",")]
        got_all = pa.Table.from_batches(batches) if batches else None
        at = 0
        for rnd, cnt in enumerate(counts):
            got = got_all.slice(at, cnt).combine_chunks() if got_all is not None else None
            at += cnt
            want = expected(world, 11 + rnd, rows, layout, [int(k) for k in keys.split(",")], r)
            assert cnt == want.num_rows
            if cnt:
                assert got.schema.types == want.schema.types
                for c in range(want.num_columns):
                    assert got.column(c).to_pylist() == want.column(c).to_pylist(), (r, rnd, want.schema.field(c).name)
            total += cnt
    assert total == 2 * rows          # every row of both rounds arrived exactly once


def test_a_peer_that_dies_mid_exchange_is_an_error_not_a_hang(host_lib, tmp_path):
    procs = launch(3, tmp_path, modes={2: "die_before_exchange"}, rows=3000, timeout_ms=8000)
    t0 = time.time()
    wait_all(procs, timeout=60)
    assert time.time() - t0 < 40
    assert procs[2].returncode == 0
    msgs = []
    for r in (0, 1):
        assert procs[r].returncode == 3
        msgs.append((tmp_path / f"r{r}.arrow.err").read_text())
        assert "closed its connection" in msgs[-1] or "is gone" in msgs[-1], msgs[-1]
    # whoever notices first names the rank that died; the other survivor may only see that first one leave
    assert any("rank 2" in m for m in msgs), msgs


def test_a_silent_peer_times_out_with_its_name(host_lib, tmp_path):
    procs = launch(2, tmp_path, modes={1: "silent"}, rows=1000, timeout_ms=1500)
    t0 = time.time()
    wait_all(procs, timeout=60)
    assert time.time() - t0 < 30
    assert procs[0].returncode == 3
    msg = (tmp_path / "r0.arrow.err").read_text()
    assert "no byte moved for 1500 ms" in msg and "rank 1" in msg, msg


def test_a_rank_that_never_shows_up_fails_the_rendezvous(host_lib, tmp_path):
    ports = ",".join(str(p) for p in free_ports(2))
    p = subprocess.Popen([sys.executable, os.path.join(HOST_DIR, "rank_main.py"), "--world", "2", "--rank", "1", "--ports", ports, "--out", str(tmp_path / "r1.arrow"),
                          "--timeout-ms", "1000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    wait_all([p], timeout=60)
    assert p.returncode == 3
    assert "did not accept a connection within 1000 ms" in (tmp_path / "r1.arrow.err").read_text()


def test_ranks_in_different_collectives_do_not_mix_their_bytes(host_lib, tmp_path):
    # rank 1 exchanges one column fewer: its validity agreement message has another size — refused by the framing, not mis-read
    procs = launch(2, tmp_path, modes={1: "wrong_collective"}, rows=2000, timeout_ms=5000)
    wait_all(procs, timeout=60)
    assert procs[0].returncode == 3 or procs[1].returncode == 3
    msgs = "".join((tmp_path / f"r{r}.arrow.err").read_text() for r in (0, 1) if (tmp_path / f"r{r}.arrow.err").exists())
    assert "the ranks disagree about the exchange" in msgs or "closed its connection" in msgs, msgs
