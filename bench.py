#!/usr/bin/env python3
"""bench.py — TPC-H SF100 Q1 stage 1 (scan → filter → project → partial 2-key hash aggregate, decimal arithmetic) on HBM-resident
Arrow columns: the configuration BASELINE.json's metric is quoted on (configs[2], 600,037,902 lineitem rows = 46.8 GB per GPU).

One "step" = one execution of the plan over the rank's lineitem shard through the C ABI (createPlan → executePlan until -1 →
releasePlan: exactly what one Spark task does).  Q1 shards by contiguous row ranges with no data-path collective:
every rank owns SF100 rows (weak scaling), rank 0 prints ONE JSON line.

  value                  whole-job rows/s at task level (createPlan..releasePlan, every launch of the task included) over the table as any
                         Arrow producer hands it over — no comet:utf8_fixed_len declaration; the declared variant is a sub-object of roofline
  roofline               SURVEY §8(d)'s algorithmic 78 B/row × the rows one task processes ÷ the time of EVERY kernel that reads those bytes:
                         the dominant kernel k_gagg (70 B/row) + the task's utf8_uniform_kernel launches (the 2 × 4 B/row of Utf8 offsets),
                         measured with HIP events on the plan's stream inside libcomet (comet_plan_kernel_stats / _aux_kernel_stats);
                         `kernels` lists each with its own ms, algorithmic bytes and PMC traffic;
                         `traffic` = HBM bytes per launch measured IN THIS RUN by two rocprofv3 passes (--pmc FETCH_SIZE / WRITE_SIZE)
                         over tools/resident.py at the same row count (null when rocprofv3 is unavailable or --no-pmc)
  cpu_baseline           the oracle's operator-at-a-time C restatement of the reference's Q1 pipeline (oracle/comet_oracle.c
                         o_q1_reference_pipeline) on the first rows of the SAME shard: one thread, and the cgroup's CPU quota of threads
  paths                  the same query through the boundary's other on-ramps on a bounded sample: host ArrowArrayStream (what the JVM
                         hands over; PCIe-bound) and NativeScan over Parquet — never `value`
  q6_sf100 / q6_sf10     extra legs: Q6 stage 1; Q6's staged loads skip most cache lines of the later columns, so its honest
                         roofline fraction is the PHYSICAL one (PMC bytes ÷ kernel time); both are printed, never a frac > 1
  q3                     BASELINE configs[3]: SF100 Q3 hash joins partitioned over the ranks (strong scaling) with per-stage ms
  q95                    BASELINE configs[4]: TPC-DS SF100 Q95 (≈16 M orders), web_sales / web_returns hash-exchanged on the order number,
                         stage A partition-local, Final on rank 0 (strong scaling); the answer is verified inside the leg against an
                         independent torch evaluation on the GPU (tpcds.q95_reference_torch, ≈ 1 s)
  cold_create_plan_ms    hiprtc compilation behind createPlan for the Q1 plan with an EMPTY code-object cache (tools/cold_plan.py)
  q6_sf10_parquet        BASELINE configs[1]: SF10 Q6 from snappy / zstd Parquet end to end (tools/parquet_q6.py)
  snappy_pipeline        the device snappy decompressor alone on 1 MiB pages (tools/snappy_bench.py)
  zstd_pipeline          the device zstd decompressor alone on 1 MiB pages (tools/snappy_bench.py --codec zstd)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SF100_ROWS = 600_037_902
SF10_ROWS = 59_986_052
HBM_PEAK_GBS = 8000.0


def cpu_quota() -> int:
    """Threads this container may keep busy: the cgroup CPU quota when there is one, else the visible CPUs."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=SF100_ROWS, help="lineitem rows per GPU (default: SF100)")
    ap.add_argument("--cpu-sample-rows", type=int, default=8_000_000, help="rows per CPU thread for the cpu_baseline legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-string-hints", action="store_true", help="do not declare the CHAR(1) key columns' fixed length in the Arrow field metadata")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--no-paths", action="store_true", help="skip the host-stream / Parquet path legs")
    ap.add_argument("--path-rows", type=int, default=20_000_000)
    ap.add_argument("--q95-orders", type=int, default=16_000_000, help="total TPC-DS orders of the q95 leg (0 = skip; SF100 = 16 M)")
    ap.add_argument("--q3-orders", type=int, default=150_000_000,
                    help="extra leg: TPC-H Q3 over all ranks, total orders rows (SF100 = 150 M, strong scaling); 0 = skip")
    ap.add_argument("--leg-timeout", type=int, default=420)
    ap.add_argument("--allow-fallback", action="store_true", help="N > 1: exit 0 even if the multi-GPU legs ran over torch's all_to_all instead of the in-library RCCL exchange")
    ap.add_argument("--no-executor-leg", action="store_true", help="skip the 8 / 16 concurrent-tasks leg")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip every extra leg (Q6, Q3, paths, PMC): headline + cpu_baseline only")
    args = ap.parse_args()

    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT compiles with the installed ROCm's compiler, as under Spark (see that module)
    import pyarrow as pa
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = init_rank(torch, dist, local_rank, world)

    from datafusion_comet_amd import native, serde as S, tpch, parallel

    plan = tpch.q1_plan()
    plan_bytes = plan.encode()
    n = args.rows
    dtab, chk = tpch.lineitem_q1_device(n, device=dev, seed=1 + rank)
    # The headline runs over the table exactly as an Arrow producer hands it over: NO producer hint.  Every task verifies on the device
    # that the two CHAR(1) key columns hold one byte per value (utf8_uniform_kernel, 4 B/row of offsets per column) before the fused kernel
    # addresses the bytes directly.  The same loop over the table whose owner DECLARES the length (field metadata comet:utf8_fixed_len,
    # verified once per buffer and cached, exec_input.cpp) is timed afterwards and reported as a sub-object, never as `value`.
    dtab_hinted = None if args.no_string_hints else dtab.with_string_hints()
    torch.cuda.synchronize()
    ncols = tpch.Q1_NUM_OUTPUT_COLS

    def step(table=None):
        # one Spark task: createPlan → executePlan until -1 → releasePlan over the rank's resident shard
        it = native.CometExecIterator([native.DeviceInput(table if table is not None else dtab, device_id=local_rank)], ncols, plan_bytes, device_id=local_rank)
        out = []
        while True:
            b = native.Native.executePlan(it.handle, ncols)
            if b is None:
                break
            out.append(b)
        stats = it.kernel_stats() + it.aux_kernel_stats()      # (k_gagg ms, launches, rows, utf8_uniform_kernel ms, launches)
        it.close()
        return out, stats

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = None
    for _ in range(args.warmup):
        result, _ = step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches, aux_ms, aux_launches = 0.0, 0, 0.0, 0
    for _ in range(args.steps):
        result, st = step()
        kernel_ms += st[0]
        launches += st[1]
        aux_ms += st[3]
        aux_launches += st[4]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the same tasks over the table WITH the owner's fixed-length declaration (first task verifies it, the rest hit the cached verdict)
    elapsed_hinted = None
    if dtab_hinted is not None:
        step(dtab_hinted)
        step(dtab_hinted)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(dtab_hinted)
        barrier()
        elapsed_hinted = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([elapsed_hinted], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed_hinted = float(tmax.item())

    # ---- outside the timed region -------------------------------------------------------------------------------------------
    # (1) every aggregate of every group of this rank's Partial states against exact torch reductions of the generating tensors
    out_tab = pa.Table.from_batches(result)
    problems = tpch.q1_check_against_torch(out_tab, chk)
    ok_local = torch.tensor([0 if problems else 1], device=dev)
    if world > 1:
        dist.all_reduce(ok_local, op=dist.ReduceOp.MIN)
    verified = bool(ok_local.item())
    # (2) the only cross-rank step of a Q1-shaped plan: gather the Partial states on rank 0, Final there
    states = parallel.gather_partial_states(out_tab, 0)
    final_rows = None
    if rank == 0 and states is not None:
        fplan = S.final_of(plan, states.schema)
        fin = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(states)], 10, fplan.encode(), device_id=local_rank))
        final_rows = sorted([[str(v) for v in r] for r in zip(*[fin.column(i).to_pylist() for i in range(fin.num_columns)])])

    # (3) CPU baseline on the first rows of the same shard, and the GPU's answer for exactly those rows as a parity check
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, dtab, plan_bytes, local_rank)

    del dtab, dtab_hinted, chk
    torch.cuda.empty_cache()

    legs = {}
    exchange = "native"
    if not args.no_extra_legs:
        if world == 1:
            legs["q6"] = run_child_leg([os.path.join(ROOT, "tools", "resident.py"), "--query", "q6", "--rows", str(args.rows), "--steps", "5"],
                                       rank, local_rank, world, args.leg_timeout, 2017)
            if args.rows != SF10_ROWS:
                legs["q6_sf10"] = run_child_leg([os.path.join(ROOT, "tools", "resident.py"), "--query", "q6", "--rows", str(SF10_ROWS), "--steps", "10"],
                                                rank, local_rank, world, args.leg_timeout, 2117)
            if not args.no_pmc:
                legs["pmc"] = measure_traffic(args, local_rank, args.rows, "q1,q6") if rank == 0 else None
                if args.rows != SF10_ROWS and rank == 0:
                    legs["pmc_sf10"] = measure_traffic(args, local_rank, SF10_ROWS, "q6")
            legs["cold_plan"] = run_child_leg([os.path.join(ROOT, "tools", "cold_plan.py"), "--query", "q1"], rank, local_rank, world, 120, 4017)
            if not args.no_paths:
                # BASELINE configs[1]: TPC-H SF10 Q6 straight from Parquet (scan + 3-predicate filter + sum), end to end, snappy and zstd
                for codec in ("snappy", "zstd"):
                    legs["pq6_" + codec] = run_child_leg([os.path.join(ROOT, "tools", "parquet_q6.py"), "--codec", codec, "--steps", "4"],
                                                         rank, local_rank, world, args.leg_timeout, 5017 + (0 if codec == "snappy" else 100))
                # … and what one Spark task sees (it owns ONE core): zstd PLAIN pages inflated on the device vs on that core
                legs["pq6_zstd_task_device"] = run_child_leg([os.path.join(ROOT, "tools", "parquet_q6.py"), "--codec", "zstd", "--steps", "3", "--scan-threads", "1"],
                                                             rank, local_rank, world, args.leg_timeout, 5217)
                legs["pq6_zstd_task_host"] = run_child_leg([os.path.join(ROOT, "tools", "parquet_q6.py"), "--codec", "zstd", "--steps", "3", "--scan-threads", "1",
                                                            "--device-decompress", "false"], rank, local_rank, world, args.leg_timeout, 5267)
                legs["snappy"] = run_child_leg([os.path.join(ROOT, "tools", "snappy_bench.py"), "--pages", "240", "--skip-one-wave"], rank, local_rank, world, 180, 5317)
                legs["zstd"] = run_child_leg([os.path.join(ROOT, "tools", "snappy_bench.py"), "--codec", "zstd", "--level", "1", "--pages", "240", "--kinds", "decimal_int64,int32_lowcard"],
                                             rank, local_rank, world, 180, 5417)
            if not args.no_paths and not args.no_executor_leg:
                # the executor's shape: 8 and 16 plans at once, one host thread and one scan thread each, one GPU, one PCIe link
                legs["executor"] = run_child_leg([os.path.join(ROOT, "tools", "executor_bench.py"), "--steps", "2", "--busy"], rank, local_rank, world, 600, 5517)
            if not args.no_paths:
                legs["paths"] = run_child_leg([os.path.join(ROOT, "tools", "paths.py"), "--query", "q1", "--rows", str(args.path_rows)],
                                              rank, local_rank, world, args.leg_timeout, 3017)
        # multi-GPU legs: the in-library exchange (RCCL send / recv groups inside libcomet.so) is probed first with a short timeout; if the
        # probe fails or hangs the legs run over torch.distributed's all_to_all instead, and say so — a leg must never stall the line
        exchange = "native"
        if world > 1 and (args.q3_orders > 0 or args.q95_orders > 0):
            probe = run_child_leg([os.path.join(ROOT, "tools", "exchange_probe.py")], rank, local_rank, world, 90, 917)
            ok_t = torch.tensor([1 if (rank != 0 or (probe is not None and probe.get("ok"))) else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if not bool(ok_t.item()):
                exchange = "torch-fallback"  # the legs report exchange_transport = "torch-fallback", never silently
            legs["exchange_probe"] = probe
        fb = ["--allow-fallback"] if exchange != "native" else []      # (the child still reports its transport; THIS process turns it into the exit code)
        if args.q3_orders > 0:
            legs["q3"] = run_child_leg([os.path.join(ROOT, "tools", "q3_dist.py"), "--orders", str(args.q3_orders), "--steps", "3", "--warmup", "1", "--kernel-times", "--exchange", exchange] + fb,
                                       rank, local_rank, world, args.leg_timeout, 1017)
        if args.q95_orders > 0:
            legs["q95"] = run_child_leg([os.path.join(ROOT, "tools", "q95_dist.py"), "--orders", str(args.q95_orders), "--steps", "2", "--warmup", "1", "--kernel-times", "--verify", "torch", "--exchange", exchange] + fb,
                                        rank, local_rank, world, args.leg_timeout, 1517)

        # HBM traffic of the two join legs on one GPU (PMC passes over the same tools, one counted run each)
        if world == 1 and rank == 0 and not args.no_pmc:
            if legs.get("q3") is not None and "roofline" in legs["q3"]:
                attach_leg_traffic(legs["q3"], measure_leg_traffic(args, local_rank, [os.path.join(ROOT, "tools", "q3_dist.py"), "--orders", str(args.q3_orders), "--steps", "1",
                                                                                      "--warmup", "1", "--no-verify"], 2))
            if legs.get("q95") is not None and "roofline" in legs["q95"]:
                attach_leg_traffic(legs["q95"], measure_leg_traffic(args, local_rank, [os.path.join(ROOT, "tools", "q95_dist.py"), "--orders", str(args.q95_orders), "--steps", "1",
                                                                                       "--warmup", "1", "--no-verify"], 2))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rows_per_s = n * world * args.steps / elapsed
        avg_kernel_ms = kernel_ms / max(launches, 1)
        # Every byte is charged to the kernel that reads it.  SURVEY §8(d)'s 78 B/row = 4 (l_shipdate) + 4·16 (decimals) + 2·(4 + 1) (two
        # Utf8 keys: offsets + the byte): k_gagg reads 70 of them — with both keys verified one byte long it addresses the bytes directly and
        # never loads an offset — and the task's utf8_uniform_kernel launches (one per key column) read the other 2·4.  `frac` is the 78 B/row
        # over ALL of those launches (the HIP-event time of k_gagg + of the verification launches, per task).
        aux_ms_per_task = aux_ms / max(args.steps, 1)
        aux_per_task = aux_launches / max(args.steps, 1)
        gagg_bytes = n * (tpch.Q1_BYTES_PER_ROW - 8)
        aux_bytes = n * 8
        algo_bytes = n * tpch.Q1_BYTES_PER_ROW            # per task: one k_gagg launch + one utf8_uniform_kernel launch per key column over the rank's n rows
        kernels_ms = avg_kernel_ms + aux_ms_per_task
        achieved = algo_bytes / (kernels_ms * 1e-3) / 1e9
        pmc = legs.get("pmc") or {}
        traffic_gagg = (pmc.get("k_gagg") or {}).get("traffic_bytes_per_launch")
        traffic_aux = (pmc.get("utf8_uniform_kernel") or {}).get("traffic_bytes_per_launch")
        traffic = (traffic_gagg + traffic_aux * aux_per_task) if (traffic_gagg and traffic_aux is not None) else traffic_gagg
        kernels = [{"name": "k_gagg", "ms": avg_kernel_ms, "launches_per_task": launches / max(args.steps, 1), "algorithmic_bytes": gagg_bytes,
                    "algorithmic_GBps": gagg_bytes / (avg_kernel_ms * 1e-3) / 1e9, "frac": gagg_bytes / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic_gagg},
                   {"name": "utf8_uniform_kernel", "ms": aux_ms_per_task, "launches_per_task": aux_per_task, "algorithmic_bytes": aux_bytes,
                    "algorithmic_GBps": (aux_bytes / (aux_ms_per_task * 1e-3) / 1e9) if aux_ms_per_task else None,
                    "frac": (aux_bytes / (aux_ms_per_task * 1e-3) / 1e9 / HBM_PEAK_GBS) if aux_ms_per_task else None,
                    "traffic": (traffic_aux * aux_per_task) if traffic_aux is not None else None,
                    "note": "ms = one HIP-event pair around the task's verification launches (they run back to back on the plan's stream)"}]
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": "k_gagg + utf8_uniform_kernel (every kernel that reads the 78 B/row)", "kernel_ms": kernels_ms,
                "launches_timed": launches, "algorithmic_bytes": algo_bytes, "kernels": kernels,
                "task_level": {"ms": ms_per_step, "algorithmic_GBps": algo_bytes / (ms_per_step * 1e-3) / 1e9,
                               "frac": algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "utf8_fixed_len_metadata": False,
                               "note": "no producer hint: every task verifies the offsets of both Utf8 key columns on the device (utf8_uniform_kernel) "
                                       "before k_gagg runs; that pass is inside ms_per_step and value"}}
        if elapsed_hinted is not None:
            ms_h = elapsed_hinted / args.steps * 1e3
            roof["task_level_with_declared_fixed_len"] = {
                "ms": ms_h, "rows_per_s": n * world * args.steps / elapsed_hinted, "frac": algo_bytes / (ms_h * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "same tasks over the same shard whose owner declares comet:utf8_fixed_len=1 on the two key columns (Arrow field metadata; "
                        "verified by utf8_uniform_kernel once per buffer, verdict cached): not `value`, no reference producer emits the key"}
        if traffic:
            roof["physical"] = {"GBps": traffic / (kernels_ms * 1e-3) / 1e9, "frac": traffic / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "bytes_per_row": traffic / n, "note": pmc.get("note")}
        line = {
            "metric": "rows/sec, TPC-H SF100 Q1 scan->filter->agg (HBM-resident Arrow columns)",
            "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i128 (Decimal128; 256-bit products) / i32 (Date32) / u8 (Utf8 keys)", "data": "synthetic",
            "config": {"workload": f"TPC-H {'SF100' if n == SF100_ROWS else 'SF10' if n == SF10_ROWS else str(n) + ' rows of'} Q1 stage 1 "
                                   "(Filter -> Project, 3 decimal ops incl. one 256-bit multiply -> partial HashAggregate, 2 Utf8 keys, 8 aggregates) per GPU",
                       "rows_per_gpu": n, "bytes_per_row_algorithmic": tpch.Q1_BYTES_PER_ROW, "parallelism": f"row-range shards x{world}",
                       "jit_toolchain": native.jit_toolchain()},
            "roofline": roof,
            "result_check": {"all_8_aggregates_of_all_groups_match_torch_on_every_rank": verified, "mismatches_rank0": problems[:4],
                             "groups_rank0": out_tab.num_rows, "final_rows_all_ranks": final_rows},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu.pop("one_thread")
            line["cpu_baseline_all_cores"] = cpu.pop("all_threads")
            line["cpu_baseline_parity"] = cpu
        paths = legs.get("paths")
        if paths is not None and "error" not in paths:
            line["paths"] = {"hbm_resident": {"rows_per_s": rows_per_s / world, "rows": n, "note": "the headline, per GPU"},
                             "sample_rows": paths["rows"], "hbm_resident_on_sample": paths["hbm_resident"],
                             "host_arrow_stream": paths["host_arrow_stream"], "parquet": paths["parquet"]}
        elif paths is not None:
            line["paths"] = paths
        for key in ("q6", "q6_sf10"):
            leg = legs.get(key)
            if leg is None:
                continue
            q = leg.get("q6", leg)
            if "error" not in q:
                tr = ((pmc if key == "q6" else (legs.get("pmc_sf10") or {})).get("k_agg") or {}).get("traffic_bytes_per_launch")
                q["roofline"] = {"bound": "hbm", "kernel": "k_agg", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_GBps": q["algorithmic_GBps"],
                                 "traffic": tr,
                                 "achieved": (tr / (q["kernel_ms"] * 1e-3) / 1e9) if tr else None,
                                 "frac": (tr / (q["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
                                 "note": "frac is PHYSICAL (PMC bytes / kernel time): lane-predicated staged loads skip most cache lines of "
                                         "the later columns, so algorithmic bytes / time exceeds the HBM peak and is not a bandwidth"}
            line["q6_sf100" if key == "q6" and args.rows == SF100_ROWS else key] = q
        if legs.get("q3") is not None:
            line["q3"] = legs["q3"]
        if legs.get("q95") is not None:
            line["q95"] = legs["q95"]
        pq6 = {c: legs.get("pq6_" + c) for c in ("snappy", "zstd") if legs.get("pq6_" + c) is not None}
        if pq6:
            line["q6_sf10_parquet"] = {c: ({"ms_best": v["sec_best"] * 1e3, "ms_median": v["sec_median"] * 1e3, "rows_per_s": v["rows_per_s"], "file_bytes": v["file_bytes"],
                                            "matches_resident_plan": v["matches_resident_plan"], "pyarrow_read_s_all_cores": v["pyarrow_read_s_all_cores"],
                                            "pages_decompressed_on_device": v.get("pages_decompressed_on_device")} if "error" not in v else v)
                                       for c, v in pq6.items()}
            # what bounds these legs: the file's bytes cross PCIe once (compressed pages cross compressed), after a copy out of the page cache;
            # peak = the link rate measured in this run (executor leg's probe), achieved = file bytes / best time; device_stage = the slowest
            # device stage's own rate on the same page shapes (the decompression pipeline, measured alone)
            link = (legs.get("executor") or {}).get("pcie_link_GBps_measured")
            for c in list(pq6):
                e = line["q6_sf10_parquet"][c]
                if "error" in e:
                    continue
                pipe = ((legs.get("snappy" if c == "snappy" else "zstd") or {}).get("decimal_int64") or {}).get("pipeline") or {}
                ach = e["file_bytes"] / (e["ms_best"] * 1e-3) / 1e9
                e["roofline"] = {"bound": "pcie", "peak": link, "unit": "GB/s", "achieved": ach, "frac": (ach / link) if link else None,
                                 "floor_ms_at_measured_link": (e["file_bytes"] / (link * 1e9) * 1e3) if link else None,
                                 "device_stage": {"name": f"{c} decompression pipeline (PLAIN decimal-as-INT64 pages)", "out_GBps": pipe.get("out_GBps"),
                                                  "kernel_ms_240_pages": pipe.get("kernel_ms")},
                                 "note": "peak = pinned -> device rate measured in this run (64 MiB copies); the file also has to leave the page cache "
                                         "(pread into pinned memory, tools/read_probe.py) before it can cross"}
            task = {k: legs.get("pq6_zstd_task_" + k) for k in ("device", "host")}
            if all(v is not None and "error" not in v for v in task.values()):
                line["q6_sf10_parquet"]["zstd_one_scan_thread"] = {k: {"ms_best": v["sec_best"] * 1e3, "pages_decompressed_on_device": v.get("pages_decompressed_on_device"),
                                                                       "matches_resident_plan": v["matches_resident_plan"]} for k, v in task.items()}
            line["q6_sf10_parquet"]["note"] = ("BASELINE configs[1] end to end (createPlan .. releasePlan over the file in the page cache): footer + page walk on the host, "
                                               "snappy pages inflated on the device (multi-kernel pipeline); zstd PLAIN pages on the device or on host threads, whichever the scan's "
                                               "thread count favours (pages_decompressed_on_device says which; zstd_one_scan_thread = what a Spark task with one core sees); "
                                               "decode + fused Q6 kernel on the device")
        if legs.get("executor") is not None:
            line["executor_shape"] = legs["executor"]
        if legs.get("snappy") is not None:
            line["snappy_pipeline"] = legs["snappy"]
        if legs.get("zstd") is not None:
            line["zstd_pipeline"] = legs["zstd"]
        if legs.get("cold_plan") is not None:
            cp = legs["cold_plan"]
            line["cold_create_plan_ms"] = cp.get("cold_create_plan_ms")
            line["cold_plan"] = cp
        if legs.get("exchange_probe") is not None:
            line["exchange_probe"] = legs["exchange_probe"]
        if pmc:
            line["pmc"] = pmc
        # the complete record (per-kernel lists, notes, PMC tables) next to the run; the printed line is its short form (< 6 KB: what the driver's record keeps)
        full_path = None
        try:
            out_dir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out_dir, exist_ok=True)
            full_path = os.path.join(out_dir, f"bench_full_n{world}.json")
            with open(full_path, "w") as f:
                json.dump(line, f)
        except OSError:
            full_path = None
        short = compact_line(line)
        short["full_record"] = os.path.relpath(full_path, ROOT) if full_path else None
        print(json.dumps(short, separators=(",", ":")))
    if world > 1:
        dist.destroy_process_group()
        rc = multi_gpu_exit_code(world, not args.no_extra_legs, args.q3_orders > 0 or args.q95_orders > 0, exchange, legs if rank == 0 else None, args.allow_fallback)
        if rc:
            sys.exit(rc)      # the multi-GPU legs did not run over the in-library RCCL exchange: the line says so, and so does the exit code


def init_rank(torch, dist, local_rank: int, world: int, backend: str = "nccl") -> str:
    """One process per GPU: the rank's device (LOCAL_RANK's) is bound BEFORE anything allocates and before a communicator exists — a process group
    created first would put its context (and every rank's first allocation) on device 0.  → the device string.  (tests/test_bench_world8_cpu.py drives this
    with recording stand-ins: eight ranks, the order of the calls.)"""
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
    return dev


def leg_env(environ, rank: int, local_rank: int, world: int, port_offset: int) -> dict:
    """The environment of a child leg: the parent's rank identity, its OWN rendezvous port (the parent's + the leg's offset: the ranks of one leg meet each other,
    never the parent's group or another leg's), none of torchrun's elastic-agent variables."""
    env = dict(environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(int(environ.get("MASTER_PORT", "29500")) + port_offset)
    env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"] = str(rank), str(local_rank), str(world)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    return env


def multi_gpu_exit_code(world: int, ran_extra_legs: bool, wanted_exchange_legs: bool, exchange: str, legs_rank0, allow_fallback: bool) -> int:
    """0, or 4 when the multi-GPU legs did not run over the in-library RCCL exchange (the probe failed and they fell back to torch's all_to_all, or a leg itself
    reported the fallback with exit code 4): the line says so in `exchange_transport`, and so does the exit code unless --allow-fallback."""
    if world <= 1:
        return 0
    fell_back = ran_extra_legs and wanted_exchange_legs and exchange != "native"
    if legs_rank0 is not None and ran_extra_legs:
        fell_back = fell_back or any((legs_rank0.get(k) or {}).get("exit_code") == 4 for k in ("q3", "q95"))
    return 4 if (fell_back and not allow_fallback) else 0


def _r(x, nd=3):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(full: dict) -> dict:
    """The ONE line the driver keeps is the last ≈ 6 KB of stdout: the headline fields, a short `roofline` and `cpu_baseline`, and every extra leg's numbers FLAT under
    `legs` — the per-kernel lists, notes and PMC tables go to the sidecar file `full_record` names."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(full[k], 4) for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {"workload": cfg.get("workload"), "rows_per_gpu": cfg.get("rows_per_gpu"), "bytes_per_row_algorithmic": cfg.get("bytes_per_row_algorithmic"),
                      "parallelism": cfg.get("parallelism"), "jit_toolchain": cfg.get("jit_toolchain")}
    ro = full.get("roofline", {})
    line["roofline"] = {"bound": ro.get("bound"), "achieved": _r(ro.get("achieved"), 1), "peak": ro.get("peak"), "unit": ro.get("unit"), "frac": _r(ro.get("frac"), 4),
                        "traffic": ro.get("traffic"), "algorithmic_bytes": ro.get("algorithmic_bytes"), "kernel_ms": _r(ro.get("kernel_ms"), 4),
                        "kernels": [{"name": k.get("name"), "ms": _r(k.get("ms"), 4), "frac": _r(k.get("frac"), 4)} for k in ro.get("kernels", [])],
                        "task_level_frac": _r((ro.get("task_level") or {}).get("frac"), 4)}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 0), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": str(cb.get("sample"))[:120]}
        ca = full.get("cpu_baseline_all_cores") or {}
        line["cpu_baseline_all_cores"] = {"value": _r(ca.get("value"), 0), "cores": ca.get("cores")}
    rc = full.get("result_check") or {}
    line["verified"] = rc.get("all_8_aggregates_of_all_groups_match_torch_on_every_rank")
    legs = {}

    def put(k, v, nd=3):
        if v is not None:
            legs[k] = _r(v, nd)

    def top_kernels(roof, n=6):
        ks = sorted((roof or {}).get("kernels") or [], key=lambda k: -(k.get("ms") or 0))[:n]
        return {k["name"]: _r(k.get("ms"), 3) for k in ks if k.get("name")}

    for name in ("q3", "q95"):
        q = full.get(name)
        if not q:
            continue
        if "error" in q:
            legs[name + "_error"] = str(q["error"])[:160]
            continue
        roof = q.get("roofline") or {}
        put(name + "_ms", q.get("sec_per_run") and q["sec_per_run"] * 1e3)
        put(name + "_frac", roof.get("frac"), 4)
        put(name + "_traffic_over_algorithmic", (roof.get("traffic") / roof["algorithmic_bytes"]) if roof.get("traffic") and roof.get("algorithmic_bytes") else None)
        put(name + "_kernel_ms_total", roof.get("kernel_ms_total_rank0"))
        put(name + "_dominant_kernel", roof.get("dominant_kernel"))
        legs[name + "_kernels_ms"] = top_kernels(roof)
        for sk, sv in (q.get("stage_ms_rank0") or {}).items():
            put(f"{name}_stage_{sk}_ms", sv)
        put(name + "_verified", q.get("verified", q.get("verified_vs_torch")))
        put(name + "_exchange_transport", q.get("exchange_transport"))
        put(name + "_n_gpus", q.get("n_gpus"))
    for key, tag in (("q6_sf100", "q6_sf100"), ("q6", "q6"), ("q6_sf10", "q6_sf10")):
        q = full.get(key)
        if q and "error" not in q:
            put(tag + "_kernel_ms", q.get("kernel_ms"), 4)
            put(tag + "_task_ms", q.get("ms_per_task"), 4)
            put(tag + "_physical_frac", (q.get("roofline") or {}).get("frac"), 4)
            put(tag + "_verified", q.get("verified_vs_torch"))
    pq = full.get("q6_sf10_parquet") or {}
    for c in ("snappy", "zstd"):
        e = pq.get(c)
        if e and "error" not in e:
            put(f"pq6_{c}_ms", e.get("ms_best"))
            put(f"pq6_{c}_ms_median", e.get("ms_median"))
            put(f"pq6_{c}_frac_of_link", (e.get("roofline") or {}).get("frac"), 4)
            put(f"pq6_{c}_floor_ms", (e.get("roofline") or {}).get("floor_ms_at_measured_link"))
            put(f"pq6_{c}_matches_resident", e.get("matches_resident_plan"))
        elif e:
            legs[f"pq6_{c}_error"] = str(e.get("error"))[:120]
    one = pq.get("zstd_one_scan_thread") or {}
    put("pq6_zstd_one_thread_device_ms", (one.get("device") or {}).get("ms_best"))
    put("pq6_zstd_one_thread_host_ms", (one.get("host") or {}).get("ms_best"))
    ex = full.get("executor_shape") or {}
    put("link_GBps", ex.get("pcie_link_GBps_measured"), 2)
    for leg, tag in (("parquet_snappy", "snappy"), ("parquet_zstd", "zstd"), ("host_stream", "host")):
        for t in ("tasks_8", "tasks_16"):
            e = ((ex.get("legs") or {}).get(leg) or {}).get(t) or {}
            put(f"exec_{tag}_{t[6:]}_frac", e.get("frac_of_measured_link"), 4)
            put(f"exec_{tag}_{t[6:]}_ms", e.get("wall_ms"))
            put(f"exec_{tag}_{t[6:]}_ms_median", e.get("wall_ms_median"))
    for key, tag in (("snappy_pipeline", "snappy"), ("zstd_pipeline", "zstd")):
        for kind, e in (full.get(key) or {}).items():
            if isinstance(e, dict) and "pipeline" in e:
                put(f"{tag}_{kind}_out_GBps", e["pipeline"].get("out_GBps"), 1)
    pt = full.get("paths") or {}
    put("host_stream_rows_per_s", (pt.get("host_arrow_stream") or {}).get("rows_per_s"), 0)
    put("parquet_path_rows_per_s", (pt.get("parquet") or {}).get("rows_per_s"), 0)
    put("cold_create_plan_ms", full.get("cold_create_plan_ms"), 1)
    tl = (ro.get("task_level_with_declared_fixed_len") or {})
    put("headline_ms_with_declared_fixed_len", tl.get("ms"))
    if full.get("exchange_probe") is not None:
        put("exchange_probe_ok", (full["exchange_probe"] or {}).get("ok"))
    line["legs"] = legs
    return line


def cpu_baseline(args, dtab, plan_bytes, local_rank):
    """The C port of the reference pipeline on host cores over the first rows of the shard (downloaded from HBM), 1 thread and the
    cgroup quota of threads; the GPU plan over the same first rows must give the same states (bit-exact)."""
    import numpy as np
    import pyarrow as pa
    from concurrent.futures import ThreadPoolExecutor
    from datafusion_comet_amd import native, tpch
    from oracle import oracle as O
    threads = cpu_quota()
    per = min(args.cpu_sample_rows, dtab.num_rows)
    m = min(per * threads, dtab.num_rows)
    sample = dtab.slice(0, m).to_arrow()
    cutoff = tpch.days(1998, 9, 2)
    O.q1_reference_pipeline(sample.slice(0, 100_000), cutoff)
    reps = 4                                       # ≥ 2 s of CPU work per leg at the ≈15 M rows/s one core reaches
    c0 = time.perf_counter()
    for _ in range(reps):
        one = O.q1_reference_pipeline(sample.slice(0, per), cutoff)
    dt1 = time.perf_counter() - c0
    slices = [sample.slice(i * per, min(per, m - i * per)) for i in range((m + per - 1) // per)]

    def work(sl):
        for _ in range(reps):
            r = O.q1_reference_pipeline(sl, cutoff)                                               # ctypes releases the GIL
        return r
    with ThreadPoolExecutor(len(slices)) as ex:
        list(ex.map(lambda sl: O.q1_reference_pipeline(sl.slice(0, 1000), cutoff), slices))      # spin the pool up
        c0 = time.perf_counter()
        parts = list(ex.map(work, slices))
        dtn = time.perf_counter() - c0
    # parity: GPU Partial states over exactly the one-thread sample
    got = pa.Table.from_batches(native.execute_to_table([native.DeviceInput(dtab.slice(0, per), device_id=local_rank)],
                                                        tpch.Q1_NUM_OUTPUT_COLS, plan_bytes, device_id=local_rank))
    same = True
    rows = {(r[0], r[1]): r for r in zip(*[got.column(i).to_pylist() for i in range(got.num_columns)])}
    if set(rows) != set(one):
        same = False
    else:
        for k, st in one.items():
            r = rows[k]
            g = [int(r[2].scaleb(2)), int(r[4].scaleb(2)), int(r[6].scaleb(4)), int(r[8].scaleb(6)), int(r[10].scaleb(2)), int(r[12].scaleb(2)),
                 int(r[14].scaleb(2))]
            same &= g == st["sums"] and [r[11], r[13], r[15], r[16]] == st["counts"]
    what = ("operator-at-a-time scalar C restatement of the reference's Q1 stage-1 pipeline (oracle/comet_oracle.c o_q1_reference_pipeline, 8192-row "
            "batches); NOT the Rust reference itself (no Rust toolchain here) and not checked to be within 2x of it on these cores")
    return {"one_thread": {"value": reps * per / dt1, "unit": "rows/s", "cores": 1, "kind": "port",
                           "sample": f"first {per} rows of the same lineitem shard x {reps} passes, {what}, {dt1:.2f} s"},
            "all_threads": {"value": reps * m / dtn, "unit": "rows/s", "cores": len(slices), "kind": "port",
                            "sample": f"first {m} rows, one contiguous {per}-row slice per thread x {reps} passes ({len(slices)} threads = the cgroup CPU quota), {dtn:.2f} s",
                            "groups": sorted("".join(k) for k in set().union(*[set(p) for p in parts]))},
            "gpu_states_equal_cpu_states_on_the_one_thread_sample": bool(same)}


def measure_leg_traffic(args, local_rank, tool_cmd, runs):
    """HBM bytes per RUN of every generated kernel (k_*) of a leg's tool: the same two rocprofv3 passes as measure_traffic over `tool_cmd`, which
    executes the query `runs` times; per kernel name the launches of all runs are summed and divided by `runs` (a probe kernel is launched
    once per join, with different sizes).  FETCH_SIZE doubled as for the headline."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"error": "rocprofv3 not found"}
    raw = {}
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    env["LOCAL_RANK"] = str(local_rank)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="comet_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable] + tool_cmd
        try:
            p = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.leg_timeout)
            if p.returncode != 0:
                return {"error": f"rocprofv3 {counter} pass exit code {p.returncode}", "log_tail": p.stdout[-400:]}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"].split("(")[0]
                    if name.startswith("k_") and r["Counter_Name"] == counter:
                        e = raw.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                        e[counter] += float(r["Counter_Value"])
                        if counter == "FETCH_SIZE":
                            e["launches"] += 1
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 {counter} pass timed out"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res = {}
    for name, e in raw.items():
        fetch, write = 2.0 * 1024.0 * e["FETCH_SIZE"] / runs, 1024.0 * e["WRITE_SIZE"] / runs
        res[name] = {"traffic_bytes_per_run": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches_per_run": e["launches"] / runs}
    return res


def attach_leg_traffic(leg, pmc):
    """per-kernel PMC traffic into a leg's roofline.kernels, their sum over ALL of the query's generated kernels into roofline.traffic"""
    if not leg or "roofline" not in leg or not pmc or "error" in pmc:
        if leg and "roofline" in leg and pmc and "error" in pmc:
            leg["roofline"]["traffic_error"] = pmc
        return
    for k in leg["roofline"]["kernels"]:
        t = pmc.get(k["name"])
        if t:
            k["traffic"] = t["traffic_bytes_per_run"]
            k["physical_GBps"] = (t["traffic_bytes_per_run"] / (k["ms"] * 1e-3) / 1e9) if k["ms"] else None
    leg["roofline"]["traffic"] = sum(t["traffic_bytes_per_run"] for t in pmc.values())
    leg["roofline"]["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the same tool with one timed run, summed over every generated kernel of the "
                                      "query, per run; FETCH_SIZE x2 (gfx950); traffic / algorithmic_bytes > 1 = re-reads and random-access line waste")


def measure_traffic(args, local_rank, rows, queries):
    """HBM bytes per launch of k_gagg (Q1) and k_agg (Q6) at this run's row count: two rocprofv3 passes (FETCH_SIZE needs 3 of the 4
    TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md) over tools/resident.py.  gfx950's FETCH_SIZE reports half the bytes of a coalesced
    streaming read, so it is doubled as that guide prescribes (calibrated there for 16 B/lane loads; tools/pmc_calibrate.py checks the
    4 and 8 B/lane loads these kernels issue — profiles/)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"error": "rocprofv3 not found"}
    res, raw = {}, {}
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    env["LOCAL_RANK"] = str(local_rank)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="comet_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "resident.py"), "--query", queries, "--rows", str(rows), "--steps", "2", "--no-check"]
        try:
            p = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.leg_timeout)
            if p.returncode != 0:
                return {"error": f"rocprofv3 {counter} pass exit code {p.returncode}", "log_tail": p.stdout[-400:]}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"].split("(")[0]
                    if name in ("k_gagg", "k_agg", "utf8_uniform_kernel") and r["Counter_Name"] == counter:
                        raw.setdefault((name, counter), []).append(float(r["Counter_Value"]))
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 {counter} pass timed out"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for name in ("k_gagg", "k_agg", "utf8_uniform_kernel"):
        f, w = raw.get((name, "FETCH_SIZE")), raw.get((name, "WRITE_SIZE"))
        if not f or not w:
            continue
        f, w = f[1:] or f, w[1:] or w           # the first launch also pages the freshly generated table in
        fetch = 2.0 * 1024.0 * sum(f) / len(f)  # KiB → B, ×2 gfx950 correction
        write = 1024.0 * sum(w) / len(w)
        res[name] = {"traffic_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches": len(f),
                     "bytes_per_row": (fetch + write) / rows}
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) in this run; FETCH_SIZE x2 per "
                   "MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)")
    res["rows"] = rows
    return res


def run_child_leg(cmd_tail, rank, local_rank, world, timeout, port_offset):
    """Run a tool as a child process of this rank (same RANK/WORLD_SIZE, its own rendezvous port); rank 0 returns the tool's
    one-line JSON (--out), every failure becomes {"error": ...}.  A hung child is killed by PID after `timeout` seconds."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"comet_leg_{os.getpid()}_{port_offset}.json")
    env = leg_env(os.environ, rank, local_rank, world, port_offset)
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
        if not os.path.exists(out):
            return {"error": f"exit code {p.returncode}", "log_tail": log[-600:]}
        with open(out) as f:
            res = json.loads(f.read())
        os.unlink(out)
        if p.returncode != 0:      # the tool wrote its line and then said no (a failed check, an exchange that fell back): keep both
            res["exit_code"] = p.returncode
            if p.returncode != 4:
                res["error"] = f"exit code {p.returncode}"
                res["log_tail"] = log[-600:]
        return res
    except Exception as e:  # an extra leg must never break the headline line
        return {"error": repr(e)} if rank == 0 else None


if __name__ == "__main__":
    main()
