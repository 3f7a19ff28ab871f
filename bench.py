#!/usr/bin/env python3
"""bench.py — TPC-H SF10 Q6 stage 1 (scan→filter→project→partial sum) on HBM-resident Arrow columns.

One "step" = one execution of the plan over the rank's lineitem shard through the C ABI
(createPlan → executePlan → releasePlan, exactly what one Spark task does).  Q6 shards by contiguous row
ranges with no data-path collective: each rank owns SF10 rows (weak scaling), rank 0
prints one JSON line.  roofline.achieved uses SURVEY §8(d)'s algorithmic bytes (52 B/row) over the
kernel time measured with HIP events on the plan's stream; cpu_baseline times the oracle's
operator-at-a-time restatement of the reference pipeline on one host core over a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SF10_ROWS = 59_986_052
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=SF10_ROWS, help="lineitem rows per GPU (default: SF10)")
    ap.add_argument("--cpu-sample-rows", type=int, default=20_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--q3-orders", type=int, default=150_000_000,
                    help="extra leg: TPC-H Q3 (hash joins + RCCL exchange) over all ranks, total orders rows (SF100 = 150 M, strong scaling); 0 = skip")
    ap.add_argument("--q3-timeout", type=int, default=420)
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the 1-GPU extra legs: TPC-DS Q95 (4 M orders, verified) and the native shuffle write")
    ap.add_argument("--no-parquet-leg", action="store_true", help="skip the extra leg: SF10 Q6 straight from a zstd Parquet file (1 GPU only)")
    ap.add_argument("--q1-rows", type=int, default=600_037_902,
                    help="extra leg: TPC-H Q1 stage 1 on HBM-resident columns, lineitem rows per GPU (SF100 = 600,037,902); 0 = skip")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    from datafusion_comet_amd import native, serde as S, tpch

    plan = tpch.q6_plan()
    plan_bytes = plan.encode()
    table = tpch.lineitem_q6(args.rows, seed=6 + rank)
    dtab = native.DeviceTable.from_arrow(table, dev)
    n = args.rows

    dinput = native.DeviceInput(dtab, device_id=local_rank)

    def step():
        # one Spark task: createPlan → executePlan until -1 → releasePlan over the rank's resident shard
        it = native.CometExecIterator([dinput.rearm()], tpch.Q6_NUM_OUTPUT_COLS, plan_bytes, device_id=local_rank)
        out = list(it_batches(it))
        stats = it.kernel_stats()
        it.close()
        return out, stats

    def it_batches(it):
        while True:
            b = native.Native.executePlan(it.handle, it.num_output_cols)
            if b is None:
                return
            yield b

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = None
    for _ in range(args.warmup):
        result, _ = step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches = 0.0, 0
    for _ in range(args.steps):
        result, st = step()
        kernel_ms += st[0]
        launches += st[1]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # outside the timed region: gather the per-rank Partial states on rank 0 and run the Final stage there
    # (the only cross-rank step of a Q6/Q1-shaped plan) — proves the N-GPU answer is the merged one
    import pyarrow as pa
    from datafusion_comet_amd import parallel
    states = parallel.gather_partial_states(pa.Table.from_batches(result) if result else None, 0)
    final_value = None
    if rank == 0 and states is not None:
        fplan = S.hash_agg(S.scan([S.decimal(35, 4), S.T_BOOL]), [], plan.aggs, S.FINAL)
        fin = native.execute_to_table([native.HostInput.from_table(states)], 1, fplan.encode(), device_id=local_rank)
        final_value = str(fin[0].column(0)[0])

    # Extra leg (reported under "q3", never part of `value`): BASELINE.json config 4, TPC-H Q3 partitioned over the ranks with
    # RCCL all-to-all exchanges (tools/q3_dist.py).  It runs in a CHILD process per rank with its own rendezvous port so that a
    # failure or hang there cannot take the Q6 line down: the child is killed by PID after --q3-timeout seconds.
    q3 = None
    if True:
        del dtab, dinput
        torch.cuda.empty_cache()
    if args.q3_orders > 0:
        q3 = run_q3_leg(args, rank, local_rank, world)
    pq = None
    if world == 1 and not args.no_parquet_leg:
        pq = run_child_leg([os.path.join(ROOT, "tools", "parquet_q6.py"), "--rows", str(args.rows), "--codec", "zstd", "--steps", "5"],
                           rank, local_rank, world, args.q3_timeout, port_offset=3017)
    q95 = shuf = None
    if world == 1 and not args.no_extra_legs:
        # BASELINE config 5 (TPC-DS Q95, here at a quarter of SF100 so that its numpy verification stays short) and §8 f1 (native shuffle write)
        q95 = run_child_leg([os.path.join(ROOT, "tools", "q95_bench.py"), "--orders", "4000000", "--reps", "2"], rank, local_rank, world, args.q3_timeout, port_offset=4017)
        shuf = run_child_leg([os.path.join(ROOT, "tools", "shuffle_bench.py"), "--rows", "20000000", "--codec", "lz4", "--reps", "2"], rank, local_rank, world,
                             args.q3_timeout, port_offset=5017)
    q1 = None
    if args.q1_rows > 0:
        q1 = run_child_leg([os.path.join(ROOT, "tools", "q1_sf100.py"), "--rows", str(args.q1_rows), "--steps", "5", "--seed", str(1 + rank)],
                           rank, local_rank, world, args.q3_timeout, port_offset=2017)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rows_per_s = n * world * args.steps / elapsed
        avg_kernel_ms = kernel_ms / max(launches, 1)
        algo_bytes = n * tpch.Q6_BYTES_PER_ROW            # per launch: one launch processes the rank's n rows
        achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # same command, gfx950 x2 FETCH correction applied); only valid for the row count it was measured on
        traffic = None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "q6_sf10_traffic.json")))
            if t["rows"] == n:
                traffic = t["traffic_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "rows/sec, TPC-H Q6 scan->filter->agg (HBM-resident Arrow columns)",
            "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i128 (Decimal128) / i32 (Date32)", "data": "synthetic",
            "config": {"workload": "TPC-H SF10 Q6 stage 1 (Filter 5 conjuncts -> Project -> partial SumDecimal) per GPU",
                       "rows_per_gpu": n, "bytes_per_row_algorithmic": tpch.Q6_BYTES_PER_ROW, "parallelism": f"row-range shards x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_agg",
                         "kernel_ms": avg_kernel_ms, "algorithmic_bytes": algo_bytes},
            "result_check": {"rank0_partial_sum": str(result[0].column(0)[0]) if result else None, "final_revenue_all_ranks": final_value},
        }
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            m = min(args.cpu_sample_rows, n)
            sample = table.slice(0, m)
            O.q6_reference_pipeline(sample.slice(0, 100_000), tpch.days(1994, 1, 1), tpch.days(1995, 1, 1), 5, 7, 2400)
            c0 = time.perf_counter()
            O.q6_reference_pipeline(sample, tpch.days(1994, 1, 1), tpch.days(1995, 1, 1), 5, 7, 2400)
            cdt = time.perf_counter() - c0
            line["cpu_baseline"] = {"value": m / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
                                    "sample": f"first {m} rows of the same lineitem shard, operator-at-a-time C restatement "
                                              f"(oracle/comet_oracle.c o_q6_reference_pipeline, 8192-row batches), {cdt:.2f} s"}
            # the same port on many host cores (one slice per thread; ctypes releases the GIL), as SURVEY §8(d) asks: 1 and all cores
            try:
                from concurrent.futures import ThreadPoolExecutor
                threads = max(1, min(os.cpu_count() or 1, 64))
                big = table                                         # the whole SF10 shard: still a bounded sample (≈ 0.5 s per core)
                per = (big.num_rows + threads - 1) // threads
                slices = [big.slice(i * per, per) for i in range(threads) if i * per < big.num_rows]
                run = lambda sl: O.q6_reference_pipeline(sl, tpch.days(1994, 1, 1), tpch.days(1995, 1, 1), 5, 7, 2400)
                with ThreadPoolExecutor(len(slices)) as ex:
                    list(ex.map(run, [sl.slice(0, 1000) for sl in slices]))          # spin the pool up
                    c0 = time.perf_counter()
                    parts = list(ex.map(run, slices))
                    cdt2 = time.perf_counter() - c0
                line["cpu_baseline_all_cores"] = {"value": big.num_rows / cdt2, "unit": "rows/s", "cores": len(slices), "kind": "port",
                                                  "sample": f"all {big.num_rows} rows, one contiguous slice per thread, {cdt2:.2f} s",
                                                  "partial_sums_add_up": str(sum(p[0] for p in parts)) == (str(result[0].column(0)[0]).replace(".", "") if result else None)}
            except Exception as e:   # never break the headline line
                line["cpu_baseline_all_cores"] = {"error": repr(e)}
        if q3 is not None:
            line["q3"] = q3
        if q1 is not None:
            line["q1_sf100_per_gpu"] = q1
        if pq is not None:
            line["q6_from_parquet"] = pq
        if q95 is not None:
            line["tpcds_q95"] = q95
        if shuf is not None:
            line["shuffle_write"] = shuf
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def run_q3_leg(args, rank, local_rank, world):
    return run_child_leg([os.path.join(ROOT, "tools", "q3_dist.py"), "--orders", str(args.q3_orders), "--steps", "3", "--warmup", "1"],
                         rank, local_rank, world, args.q3_timeout, port_offset=1017)


def run_child_leg(cmd_tail, rank, local_rank, world, timeout, port_offset):
    """Run a tool as a child process of this rank (same RANK/WORLD_SIZE, its own rendezvous port); rank 0 returns the tool's
    one-line JSON (--out), every failure becomes {"error": ...}.  A hung child is killed by PID after `timeout` seconds."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"comet_leg_{os.getpid()}_{port_offset}.json")
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"] = str(rank), str(local_rank), str(world)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    cmd = [sys.executable] + list(cmd_tail) + ["--out", out]
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            log, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            return {"error": f"timed out after {timeout} s"} if rank == 0 else None
        if rank != 0:
            return None
        if p.returncode != 0 or not os.path.exists(out):
            return {"error": f"exit code {p.returncode}", "log_tail": log[-600:]}
        with open(out) as f:
            res = json.loads(f.read())
        os.unlink(out)
        return res
    except Exception as e:  # an extra leg must never break the headline line
        return {"error": repr(e)} if rank == 0 else None


if __name__ == "__main__":
    main()
