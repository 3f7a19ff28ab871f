#!/usr/bin/env python3
"""The shape Spark actually runs: MANY plan handles at once on one GPU, one host thread each (jni_api.rs:133-170 — createPlan / executePlan /
releasePlan of one handle come from one task thread, an executor runs as many task threads as it has cores), every task with ONE scan
thread (spark.comet.gpu.scanThreads=1: a task owns one core), all of them sharing one PCIe link.

Legs, each with T = 8 and 16 concurrent tasks over one input split T ways:
  parquet_snappy / parquet_zstd   TPC-H SF10 Q6 from Parquet: task t scans the t-th byte range of the file (row groups by the midpoint rule),
                                  filters and emits its partial sum; the partial sums are added up and compared with one plan over the whole table
  host_stream                     the same query over host ArrowArrayStreams (8192-row batches, what CometBatchIterator hands over): task t gets
                                  the t-th slice of the rows
Reported per leg: wall time of the whole wave of tasks (all started together, until the last one has released its plan), aggregate rows/s,
bytes that crossed PCIe per second (file bytes for Parquet — compressed pages cross compressed —, Arrow bytes for the stream) against the
link rate measured in the same run (pinned → device, 64 MiB copies), and — with --busy — the fraction of that wall time during which at
least one kernel was running (rocprofv3 kernel trace of a second run of the same leg).  One JSON line."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
import threading
import time
from decimal import Decimal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def link_rate(torch, nbytes=1 << 30):
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for off in range(0, nbytes, 64 << 20):
            dev[off:off + (64 << 20)].copy_(host[off:off + (64 << 20)], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return nbytes / best / 1e9


def cfs_throttled_us():
    """microseconds this cgroup has spent throttled by its CPU quota so far (cgroup v2 cpu.stat; None when there is no such file)"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1])
    except OSError:
        pass
    return None


def run_wave(make_task, T, native, ncols):
    """T tasks, one thread each; → (wall seconds, [result batches per task])."""
    results = [None] * T
    errors = []
    go = threading.Barrier(T + 1)

    def work(t):
        try:
            inputs, plan, conf = make_task(t)
            go.wait()
            h = native.Native.createPlan(inputs, plan, conf, 1, 8192, 0)
            out = []
            while True:
                b = native.Native.executePlan(h, ncols)
                if b is None:
                    break
                out.append(b)
            native.Native.releasePlan(h)
            results[t] = out
        except Exception as e:   # noqa: BLE001 — reported, the leg fails
            errors.append(repr(e))
            try:
                go.abort()
            except Exception:   # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    go.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    if errors:
        raise RuntimeError("; ".join(errors[:3]))
    return wall, results


def q6_total(results):
    """sum of the tasks' partial sums (Q6 stage 1 emits (sum decimal, is_empty) per task)"""
    total = Decimal(0)
    for out in results:
        for b in out:
            v = b.column(0)[0].as_py()
            if v is not None:
                total += v
    return total


def busy_fraction(argv_tail):
    """kernel-busy fraction of the last wave of a profiled re-run of one leg"""
    d = tempfile.mkdtemp(prefix="execbusy")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable, os.path.abspath(__file__)] + argv_tail + ["--steps", "1", "--no-link"]
    try:
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, cwd="/tmp", check=False)
        files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        if not files:
            return None
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(files[0])))
        if not ev:
            return None
        # the last burst (events less than 20 ms apart) is the timed wave
        burst = [ev[-1]]
        for s, e in reversed(ev[:-1]):
            if burst[-1][0] - e > 20_000_000:
                break
            burst.append((s, e))
        burst.reverse()
        span = max(e for _, e in burst) - burst[0][0]
        busy, cur_s, cur_e = 0, burst[0][0], burst[0][1]
        for s, e in burst[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        return busy / span if span else None
    except Exception:   # noqa: BLE001
        return None
    finally:
        subprocess.run(["rm", "-rf", d], check=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", default="8,16")
    ap.add_argument("--legs", default="parquet_snappy,parquet_zstd,host_stream")
    ap.add_argument("--rows", type=int, default=59_986_052, help="lineitem rows of the Parquet legs (SF10)")
    ap.add_argument("--stream-rows", type=int, default=32_000_000, help="rows of the host-stream leg (52 B/row of Arrow data)")
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--busy", action="store_true", help="also measure the kernel-busy fraction (one more, profiled, run per leg and task count)")
    ap.add_argument("--no-link", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import pyarrow.parquet as papq
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    from datafusion_comet_amd import native, serde as S, tpch
    res = {"what": "concurrent tasks on one GPU, one host thread and one scan thread each (the executor's shape)", "legs": {}}
    link = None if a.no_link else link_rate(torch)
    if link:
        res["pcie_link_GBps_measured"] = link
    conf = S.config_map({"spark.comet.gpu.scanThreads": "1"})
    ncols = tpch.Q6_NUM_OUTPUT_COLS
    ok_all = True
    for leg in a.legs.split(","):
        if leg.startswith("parquet_"):
            codec = leg.split("_", 1)[1]
            path = os.path.join(a.dir, f"lineitem_q6_{a.rows}_{codec}.parquet")
            table = tpch.lineitem_q6(a.rows, seed=6)
            if not os.path.exists(path):
                papq.write_table(table, path, row_group_size=1 << 20, compression=codec, use_dictionary=True, store_decimal_as_integer=True, data_page_size=1 << 20)
            fsize = os.path.getsize(path)
            names, types = table.schema.names, [tpch.DEC, tpch.DEC, tpch.DEC, S.T_DATE]
            whole = native.execute_to_table([], ncols, tpch.q6_plan(source=S.native_scan([path], names, types)).encode())
            want = q6_total([whole])
            nrows, moved = a.rows, fsize
            del table

            def make_task(t, T, path=path, fsize=fsize, names=names, types=types):
                lo, hi = fsize * t // T, fsize * (t + 1) // T
                return [], tpch.q6_plan(source=S.native_scan([(path, lo, hi - lo, fsize)], names, types)).encode(), conf
        else:
            table = tpch.lineitem_q6(a.stream_rows, seed=6)
            whole = native.execute_to_table([native.HostInput.from_table(table, 8192)], ncols, tpch.q6_plan().encode())
            want = q6_total([whole])
            nrows, moved = a.stream_rows, a.stream_rows * tpch.Q6_BYTES_PER_ROW
            plan = tpch.q6_plan().encode()

            def make_task(t, T, table=table, plan=plan):
                lo, hi = table.num_rows * t // T, table.num_rows * (t + 1) // T
                return [native.HostInput.from_table(table.slice(lo, hi - lo), 8192)], plan, conf
        res["legs"][leg] = {"rows": nrows, "bytes_over_pcie": moved}
        for T in (int(x) for x in a.tasks.split(",")):
            best, ok, walls, throttled = None, True, [], []
            for it in range(a.steps + 1):
                time.sleep(0.06)      # (outside the timed wave: a profile of this run tells the waves apart by the silence between them)
                thr0 = cfs_throttled_us()
                wall, results = run_wave(lambda t: make_task(t, T), T, native, ncols)
                thr1 = cfs_throttled_us()
                if it and thr0 is not None and thr1 is not None:
                    throttled.append(round((thr1 - thr0) / 1e3, 2))
                ok = ok and q6_total(results) == want
                if it:
                    best = wall if best is None else min(best, wall)
                    walls.append(wall)
            if best is None:
                best = wall
            entry = {"wall_ms": best * 1e3, "wall_ms_median": sorted(walls)[len(walls) // 2] * 1e3 if walls else best * 1e3, "wall_ms_all": [round(w * 1e3, 2) for w in walls], "rows_per_s": nrows / best, "pcie_GBps": moved / best / 1e9, "answers_match_one_plan": ok}
            if throttled:
                entry["cfs_throttled_ms_all"] = throttled      # time the cgroup's CPU quota held the process's threads during each timed wave
            if link:
                entry["frac_of_measured_link"] = moved / best / 1e9 / link
            if a.busy:
                entry["gpu_busy_fraction"] = busy_fraction(["--legs", leg, "--tasks", str(T), "--rows", str(a.rows), "--stream-rows", str(a.stream_rows), "--dir", a.dir])
            res["legs"][leg][f"tasks_{T}"] = entry
            ok_all = ok_all and ok
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    sys.exit(0 if ok_all else 3)


if __name__ == "__main__":
    main()
