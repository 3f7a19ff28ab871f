"""The RCCL wire of the hash exchange, driven on a CPU-only box ("reduce what can go wrong on first contact").

csrc/exchange_rccl.hpp — the librccl loader and RcclTransportT, the text libcomet.so compiles over HBM — is instantiated here over host memory
(tests/exchange_host/) and answered by a stand-in librccl.so (tests/fake_rccl/, named through COMET_RCCL_LIBRARY) that moves the bytes between
PROCESSES over TCP and is strict about what RCCL is strict about: ncclDataType_t values turned into byte counts (a send / recv pair whose sizes
differ is an error), group semantics (a send outside a group blocks until it is received), a rank's message to itself, ncclAllGather's
per-rank element count.  2 and 8 processes run csrc/exchange_core.hpp's orchestration over it; every rank's result is compared with the
oracle's Spark partitioning, the communicator reports N ranks (comet_comm_stats's ncclCommCount), and the stand-in's own log of the calls
matches what the transport promises: ONE allgather per split, ONE group per buffer, one send and one recv per non-empty slice.
Test infrastructure only: nothing here is loaded by the product."""
import json
import os
import subprocess
import sys

import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "tests", "exchange_host")
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
sys.path.insert(0, HOST_DIR)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_native_exchange_procs_cpu import expected, wait_all      # noqa: E402


@pytest.fixture(scope="module")
def shim():
    os.makedirs(os.path.join(HOST_DIR, "_build"), exist_ok=True)
    os.makedirs(os.path.join(FAKE_DIR, "_build"), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", os.path.join(HOST_DIR, "host_exchange.cpp"), "-o",
                           os.path.join(HOST_DIR, "_build", "libcomet_exchange_host.so"), "-ldl"])
    fake = os.path.join(FAKE_DIR, "_build", "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-pthread", os.path.join(FAKE_DIR, "fake_rccl.c"), "-o", fake])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return fake


def launch(world, tmp_path, fake, **kw):
    env = dict(os.environ, COMET_RCCL_LIBRARY=fake)
    procs = []
    for r in range(world):
        cmd = [sys.executable, os.path.join(HOST_DIR, "rank_main.py"), "--world", str(world), "--rank", str(r), "--ports", "0", "--wire", "rccl",
               "--id-file", str(tmp_path / "unique_id"), "--out", str(tmp_path / f"r{r}.arrow")]
        for k, v in kw.items():
            cmd += ["--" + k.replace("_", "-"), str(v)]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    return procs


@pytest.mark.parametrize("world,layout,rows,keys", [(2, "even", 5000, "0,4"), (8, "even", 16001, "0,4"), (8, "hole", 4000, "4"), (8, "even", 5, "0")])
def test_rccl_transport_call_sequence_in_n_processes(shim, tmp_path, world, layout, rows, keys):
    rounds = 2
    procs = launch(world, tmp_path, shim, seed=21, rows=rows, layout=layout, keys=keys, rounds=rounds)
    logs = wait_all(procs, timeout=240)
    for r, p in enumerate(procs):
        err = tmp_path / f"r{r}.arrow.err"
        assert p.returncode == 0, (logs[r][-1200:], err.read_text() if err.exists() else "")
    total, sent, received = 0, 0, 0
    for r in range(world):
        with pa.ipc.open_file(str(tmp_path / f"r{r}.arrow")) as f:
            batches = [f.get_batch(i) for i in range(f.num_record_batches)]
        counts = [int(x) for x in (tmp_path / f"r{r}.arrow.rows").read_text().split(",")]
        got_all = pa.Table.from_batches(batches) if batches else None
        at = 0
        for rnd, cnt in enumerate(counts):
            got = got_all.slice(at, cnt).combine_chunks() if got_all is not None else None
            at += cnt
            want = expected(world, 21 + rnd, rows, layout, [int(k) for k in keys.split(",")], r)
            assert cnt == want.num_rows
            if cnt:
                assert got.schema.types == want.schema.types
                for c in range(want.num_columns):
                    assert got.column(c).to_pylist() == want.column(c).to_pylist(), (r, rnd, want.schema.field(c).name)
            total += cnt
        st = json.loads((tmp_path / f"r{r}.arrow.stats").read_text())
        # what the wire says about itself: N ranks, this rank's number (comet_comm_stats → ncclCommCount / ncclCommUserRank)
        assert st["comm_count"] == world and st["comm_rank"] == r
        w = st["wire"]
        assert w is not None
        # the transport's own byte accounting equals what the wire moved to / from OTHER ranks
        assert (st["bytes_sent"], st["bytes_received"]) == (w["bytes_out"], w["bytes_in"])
        # ONE group per buffer, and inside it at most one send and one recv per peer (a slice of zero units is neither sent nor posted)
        assert w["groups"] > 0 and w["sends"] <= w["groups"] * world and w["recvs"] <= w["groups"] * world
        # ONE allgather per count exchange (exchange_core.hpp:75,172,213): the row split, the validity-on-any-rank agreement, one byte split per
        # Utf8 / Binary column (the table has two) — per round
        assert w["allgathers"] == rounds * 4
        sent += st["bytes_sent"]
        received += st["bytes_received"]
    assert total == rounds * rows          # every row of both rounds arrived exactly once
    assert sent == received                # and every byte that left a rank arrived at another


def test_a_mismatched_pair_is_an_error_the_stand_in_reports(shim, tmp_path):
    """the stand-in is strict: rank 1 runs a different collective (one column fewer) → its slices no longer pair up with the peers' → an error on some
    rank, never a silent pass or a hang (what a wrong count / datatype in RcclTransportT would look like on first contact)"""
    env = dict(os.environ, COMET_RCCL_LIBRARY=shim)
    procs = []
    for r in range(2):
        cmd = [sys.executable, os.path.join(HOST_DIR, "rank_main.py"), "--world", "2", "--rank", str(r), "--ports", "0", "--wire", "rccl", "--id-file", str(tmp_path / "unique_id"),
               "--out", str(tmp_path / f"r{r}.arrow"), "--rows", "2000", "--keys", "0"] + (["--mode", "wrong_collective"] if r == 1 else [])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    wait_all(procs, timeout=200)
    errs = [(tmp_path / f"r{r}.arrow.err").read_text() for r in range(2) if (tmp_path / f"r{r}.arrow.err").exists()]
    assert any(p.returncode != 0 for p in procs) and errs, "a mismatched collective passed silently"
    assert any("exchange" in e for e in errs)
