#!/usr/bin/env python3
"""TPC-H Q6 straight from Parquet (BASELINE.json config 2: "Parquet scan + 3-predicate filter + sum agg"):
lineitem's four Q6 columns written by pyarrow the way Spark writes them (decimal(12,2) as INT64, dictionary on, 1 M-row
row groups), scanned by the NativeScan plan (host: footer / page headers / decompression; device: level, dictionary and
value decode), filtered and aggregated by the fused Q6 kernel.  Prints one JSON line with the end-to-end time, the
decode kernels' share (HIP events via comet_plan_kernel_stats are for the aggregate only; use rocprofv3 for the split)
and a pyarrow.parquet read of the same file on the host cores as the CPU reference."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=59_986_052)
    ap.add_argument("--codec", default="zstd", choices=["zstd", "snappy", "none"])
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-dictionary", action="store_true")
    ap.add_argument("--clustered", action="store_true", help="rows in l_shipdate order (a table clustered by date), written with page indexes, and Q6's date range pushed into the scan as data filters: "
                    "most pages are pruned, the pages at the range's ends are kept in pieces")
    ap.add_argument("--out", default="", help="also write the JSON line here")
    ap.add_argument("--scan-threads", type=int, default=0, help="host threads the scan may use (spark.comet.gpu.scanThreads; 1 = what a Spark task owns)")
    ap.add_argument("--device-decompress", default="auto", choices=["auto", "true", "false"], help="spark.comet.gpu.scan.deviceDecompress")
    a = ap.parse_args()
    import pyarrow as pa
    import pyarrow.parquet as papq
    from datafusion_comet_amd import native, serde as S, tpch
    os.makedirs(a.dir, exist_ok=True)
    path = os.path.join(a.dir, f"lineitem_q6_{a.rows}_{a.codec}{'_plain' if a.no_dictionary else ''}{'_clustered' if a.clustered else ''}.parquet")
    table = tpch.lineitem_q6(a.rows, seed=6)
    if a.clustered:
        import pyarrow.compute as pc
        table = table.take(pc.sort_indices(table, sort_keys=[(table.schema.names[3], "ascending")]))
    if not os.path.exists(path):
        t0 = time.perf_counter()
        papq.write_table(table, path, row_group_size=1 << 20, compression=None if a.codec == "none" else a.codec,
                         use_dictionary=not a.no_dictionary, store_decimal_as_integer=True, data_page_size=(64 << 10) if a.clustered else (1 << 20),
                         write_page_index=a.clustered)
        print(f"wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB in {time.perf_counter() - t0:.1f} s", file=sys.stderr)
    fsize = os.path.getsize(path)
    filters = []
    if a.clustered:
        ship = S.col(3, S.T_DATE)
        filters = [S.gt_eq(ship, S.lit(tpch.days(1994, 1, 1), S.T_DATE)), S.lt(ship, S.lit(tpch.days(1995, 1, 1), S.T_DATE))]
    src = S.native_scan([path], table.schema.names, [tpch.DEC, tpch.DEC, tpch.DEC, S.T_DATE], data_filters=filters)
    plan = tpch.q6_plan(source=src).encode()
    cfg = {"spark.comet.gpu.scan.deviceDecompress": a.device_decompress}
    if a.scan_threads:
        cfg["spark.comet.gpu.scanThreads"] = str(a.scan_threads)
    conf = S.config_map(cfg)
    want = None
    times = []
    on_device = None
    for it in range(a.steps + 1):
        t0 = time.perf_counter()
        h = native.Native.createPlan([], plan, conf, 1, 8192, 0)
        t1 = time.perf_counter()
        out = [native.Native.executePlan(h, tpch.Q6_NUM_OUTPUT_COLS)]
        t2 = time.perf_counter()
        assert native.Native.executePlan(h, tpch.Q6_NUM_OUTPUT_COLS) is None
        if on_device is None:
            try:
                lib = native.lib()
                n = lib.comet_plan_metrics(h, None, 0)
                buf = ctypes.create_string_buffer(max(int(n), 1))
                lib.comet_plan_metrics(h, buf, n)
                m = S.decode_metric_node(buf.raw[:n])
                while m and m[1]:
                    m = m[1][0]
                on_device = m[0].get("pages_decompressed_on_device") if m else None
            except Exception:
                on_device = -1
        native.Native.releasePlan(h)
        dt = time.perf_counter() - t0
        if os.environ.get("COMET_TRACE_STAGES"):
            print(f"[tool] createPlan {1e3 * (t1 - t0):.2f} ms, executePlan {1e3 * (t2 - t1):.2f} ms, eof+release {1e3 * (time.perf_counter() - t2):.2f} ms", file=sys.stderr)
        if it:
            times.append(dt)
        got = str(out[0].column(0)[0])
        want = want or got
        assert got == want
    # the same plan over HBM-resident columns gives the expected value
    ref = native.execute_to_table([native.DeviceInput(native.DeviceTable.from_arrow(table))], tpch.Q6_NUM_OUTPUT_COLS, tpch.q6_plan().encode())
    ok = str(ref[0].column(0)[0]) == want
    # CPU reference decoder on the host cores
    t0 = time.perf_counter()
    papq.read_table(path)
    cpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    papq.read_table(path, use_threads=False)
    cpu1 = time.perf_counter() - t0
    best = min(times)
    decoded = a.rows * tpch.Q6_BYTES_PER_ROW
    line = json.dumps({"query": "tpch_q6_parquet", "rows": a.rows, "codec": a.codec, "dictionary": not a.no_dictionary, "file_bytes": fsize,
                      "sec_best": best, "sec_median": sorted(times)[len(times) // 2], "sec_all": [round(t, 4) for t in times], "rows_per_s": a.rows / best,
                      "encoded_GBps": fsize / best / 1e9, "decoded_arrow_GBps": decoded / best / 1e9, "result": want, "matches_resident_plan": ok,
                      "pyarrow_read_s_all_cores": cpu, "pyarrow_read_s_1_core": cpu1, "host_cores": os.cpu_count(),
                      "scan_threads": a.scan_threads or None, "device_decompress": a.device_decompress, "pages_decompressed_on_device": on_device})
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
