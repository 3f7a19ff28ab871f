#!/usr/bin/env python3
"""The JVM-shaped path: inputs arrive as host Arrow batches through the C Stream interface (what CometNativeArrowSource
exports), are staged through pinned memory and cross PCIe.  Measures TPC-H Q6 (SF10 by default) end to end for a few batch
sizes, and BASELINE config 1 (1 M rows int64/float64 Filter+Project).  The rate here is PCIe/host bound and is never bench.py's
`value`."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=59_986_052)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import numpy as np
    import pyarrow as pa
    from datafusion_comet_amd import native, serde as S, tpch
    table = tpch.lineitem_q6(a.rows, seed=6)
    plan = tpch.q6_plan().encode()
    res = {"rows": a.rows, "bytes": a.rows * tpch.Q6_BYTES_PER_ROW, "q6": {}}
    for batch_rows in (8192, 131072, 4 << 20):
        times = []
        for it in range(a.steps + 1):
            inp = native.HostInput.from_table(table, batch_rows)
            t0 = time.perf_counter()
            out = native.execute_to_table([inp], tpch.Q6_NUM_OUTPUT_COLS, plan)
            dt = time.perf_counter() - t0
            if it:
                times.append(dt)
        best = min(times)
        res["q6"][str(batch_rows)] = {"sec": best, "rows_per_s": a.rows / best, "GBps": a.rows * tpch.Q6_BYTES_PER_ROW / best / 1e9,
                                      "result": str(out[0].column(0)[0])}
    # config 1: Scan → Filter(a < 500000 AND b IS NOT NULL) → Projection(a + 1, b * 2.0, a), 1 M rows
    rng = np.random.default_rng(42)
    n = 1_000_000
    t1 = pa.table({"a": pa.array(rng.integers(0, 1_000_000, n), pa.int64()), "b": pa.array(rng.random(n))})
    ca, cb = S.col(0, S.T_INT64), S.col(1, S.T_DOUBLE)
    p1 = S.project(S.filter_(S.scan([S.T_INT64, S.T_DOUBLE]), S.and_(S.lt(ca, S.lit(500000, S.T_INT64)), S.is_not_null(cb))),
                   [S.math("add", ca, S.lit(1, S.T_INT64), S.T_INT64), S.math("multiply", cb, S.lit(2.0, S.T_DOUBLE), S.T_DOUBLE), ca]).encode()
    times = []
    for it in range(6):
        inp = native.HostInput.from_table(t1, 8192)
        t0 = time.perf_counter()
        out = native.execute_to_table([inp], 3, p1, batch_size=8192)
        dt = time.perf_counter() - t0
        if it:
            times.append(dt)
    res["config1"] = {"sec": min(times), "rows_per_s": n / min(times), "out_rows": sum(b.num_rows for b in out), "out_batches": len(out)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
