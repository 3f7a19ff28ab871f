#!/usr/bin/env python3
"""TPC-H Q1 / Q6 stage 1 over HBM-resident lineitem columns on ONE MI355X, through the C ABI (createPlan → executePlan →
releasePlan per step): the table is generated in HBM with torch (SF100 = 600,037,902 rows: Q1 46.8 GB, Q6 31.2 GB), the result is
checked against independent exact torch reductions of the generating tensors (Q1: all eight aggregates of all four groups).
Prints one JSON object per query; --out writes {"q1": {...}, "q6": {...}}.  Also the command the rocprofv3 passes profile
(bench.py's `roofline.traffic`, profiles/)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_query(q, rows, steps, seed, local, check):
    import pyarrow as pa
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    from datafusion_comet_amd import native, tpch
    dev = f"cuda:{local}"
    if q == "q1":
        dt, chk = tpch.lineitem_q1_device(rows, device=dev, seed=seed)
        plan, ncols, bpr, kernel = tpch.q1_plan(), tpch.Q1_NUM_OUTPUT_COLS, tpch.Q1_BYTES_PER_ROW, "k_gagg"
    else:
        dt, chk = tpch.lineitem_q6_device(rows, device=dev, seed=seed)
        plan, ncols, bpr, kernel = tpch.q6_plan(), tpch.Q6_NUM_OUTPUT_COLS, tpch.Q6_BYTES_PER_ROW, "k_agg"
    torch.cuda.synchronize()
    pb = plan.encode()
    ms, wall, out = [], [], None
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        it = native.CometExecIterator([native.DeviceInput(dt, device_id=local)], ncols, pb, device_id=local)
        batches = []
        while True:
            b = native.Native.executePlan(it.handle, ncols)
            if b is None:
                break
            batches.append(b)
        out = pa.Table.from_batches(batches)
        st = it.kernel_stats()
        ms.append(st[0] / max(st[1], 1))
        it.close()
        wall.append((time.perf_counter() - t0) * 1e3)
    k = sum(ms[1:]) / len(ms[1:])
    w = min(wall[1:])
    ok = None
    if check:
        if q == "q1":
            problems = tpch.q1_check_against_torch(out, chk)
            for pr in problems:
                print("  MISMATCH", pr, file=sys.stderr)
            ok = not problems
        else:
            ok = int(out.column(0)[0].as_py().scaleb(4)) == tpch.q6_torch_reference(chk)
    res = {"query": f"tpch_{q}_stage1", "rows": rows, "kernel": kernel, "kernel_ms": k, "kernel_ms_min": min(ms[1:]), "ms_per_task": w,
           "rows_per_s": rows / w * 1e3, "bytes_per_row_algorithmic": bpr,
           "algorithmic_GBps": rows * bpr / k / 1e6, "algorithmic_frac_of_8TBps": rows * bpr / k / 1e6 / 8000.0,
           "verified_vs_torch": ok, "groups": out.num_rows}
    del dt, chk
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--query", default="q1", help="q1, q6 or q1,q6")
    ap.add_argument("--rows", type=int, default=600_037_902)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--out", default="", help="write a one-line JSON summary here")
    args = ap.parse_args()
    import torch
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    res = {}
    for q in args.query.split(","):
        res[q] = run_query(q, args.rows, args.steps, args.seed, local, not args.no_check)
        print(json.dumps(res[q]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(res) + "\n")
    sys.exit(0 if all(r["verified_vs_torch"] in (True, None) for r in res.values()) else 1)


if __name__ == "__main__":
    main()
