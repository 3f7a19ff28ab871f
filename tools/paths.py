#!/usr/bin/env python3
"""The same query through the three on-ramps of the boundary, on a bounded sample (default 20 M lineitem rows):
  hbm_resident        Arrow C *Device* stream, columns already in HBM (what bench.py's headline times at full size)
  host_arrow_stream   host ArrowArrayStream of 8192-row batches — what the JVM hands over (CometBatchIterator); staged through
                      pinned memory and carried over PCIe, so this is the rate a Spark executor sees
  parquet             NativeScan over a Parquet file written the way Spark writes it (decimal(12,2) as INT64, dictionary pages,
                      1 Mi-row row groups): host footer / page walk, device decode
Each is createPlan → executePlan until -1 → releasePlan, best of --steps, results compared with each other.  One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--query", default="q1", choices=["q1", "q6"])
    ap.add_argument("--rows", type=int, default=20_000_000)
    ap.add_argument("--codec", default="snappy", choices=["zstd", "snappy", "none"])
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import pyarrow as pa
    import pyarrow.parquet as papq
    from datafusion_comet_amd import native, serde as S, tpch
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.query == "q1":
        table = tpch.lineitem_q1(a.rows, seed=1)
        mk, ncols, bpr = tpch.q1_plan, tpch.Q1_NUM_OUTPUT_COLS, tpch.Q1_BYTES_PER_ROW
        types = [tpch.DEC] * 4 + [S.T_STRING, S.T_STRING, S.T_DATE]
    else:
        table = tpch.lineitem_q6(a.rows, seed=6)
        mk, ncols, bpr = tpch.q6_plan, tpch.Q6_NUM_OUTPUT_COLS, tpch.Q6_BYTES_PER_ROW
        types = [tpch.DEC] * 3 + [S.T_DATE]
    plan = mk().encode()

    def canon(batches):
        t = pa.Table.from_batches(batches)
        rows = list(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]))
        return sorted(rows, key=repr)

    def best_of(make_inputs, plan_bytes):
        times, out = [], None
        for it in range(a.steps + 1):
            inputs = make_inputs()
            t0 = time.perf_counter()
            out = native.execute_to_table(inputs, ncols, plan_bytes, device_id=local)
            dt = time.perf_counter() - t0
            if it:
                times.append(dt)
        return min(times), canon(out)

    res = {"query": f"tpch_{a.query}_stage1", "rows": a.rows, "arrow_bytes": a.rows * bpr}
    dtab = native.DeviceTable.from_arrow(table, f"cuda:{local}")
    sec, want = best_of(lambda: [native.DeviceInput(dtab, device_id=local)], plan)
    res["hbm_resident"] = {"sec": sec, "rows_per_s": a.rows / sec}
    del dtab
    sec, got = best_of(lambda: [native.HostInput.from_table(table, 8192)], plan)
    res["host_arrow_stream"] = {"sec": sec, "rows_per_s": a.rows / sec, "arrow_GBps": a.rows * bpr / sec / 1e9, "batch_rows": 8192,
                                "matches_resident": got == want}
    path = os.path.join(a.dir, f"lineitem_{a.query}_{a.rows}_{a.codec}.parquet")
    if not os.path.exists(path):
        papq.write_table(table, path, row_group_size=1 << 20, compression=None if a.codec == "none" else a.codec, use_dictionary=True,
                         store_decimal_as_integer=True, data_page_size=1 << 20)
    fsize = os.path.getsize(path)
    pplan = mk(source=S.native_scan([path], table.schema.names, types)).encode()
    sec, got = best_of(lambda: [], pplan)
    res["parquet"] = {"sec": sec, "rows_per_s": a.rows / sec, "codec": a.codec, "file_bytes": fsize, "file_GBps": fsize / sec / 1e9,
                      "decoded_arrow_GBps": a.rows * bpr / sec / 1e9, "matches_resident": got == want}
    t0 = time.perf_counter()
    papq.read_table(path)
    res["parquet"]["pyarrow_read_s_all_cores"] = time.perf_counter() - t0
    try:
        os.unlink(path)
    except OSError:
        pass
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    sys.exit(0 if res["host_arrow_stream"]["matches_resident"] and res["parquet"]["matches_resident"] else 3)


if __name__ == "__main__":
    main()
