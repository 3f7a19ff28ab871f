#!/usr/bin/env python3
"""Kernel-level numbers for DESIGN.md: runs Q6 / Q1 stage 1 on HBM-resident columns and prints kernel time
(HIP events inside libcomet), algorithmic GB/s and the host-side cost of createPlan/executePlan/releasePlan."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q6-rows", type=int, default=59_986_052)
    ap.add_argument("--q1-rows", type=int, default=50_000_000)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    from datafusion_comet_amd import native, tpch

    for name, rows, mk, plan, ncols, bpr in (("q6", args.q6_rows, tpch.lineitem_q6, tpch.q6_plan(), tpch.Q6_NUM_OUTPUT_COLS, tpch.Q6_BYTES_PER_ROW),
                                             ("q1", args.q1_rows, tpch.lineitem_q1, tpch.q1_plan(), tpch.Q1_NUM_OUTPUT_COLS, tpch.Q1_BYTES_PER_ROW)):
        if rows <= 0:
            continue
        t0 = time.perf_counter()
        table = mk(rows)
        dt = native.DeviceTable.from_arrow(table, "cuda:0")
        del table
        gen_s = time.perf_counter() - t0
        pb = plan.encode()
        tc = te = tr = 0.0
        kms = 0.0
        for i in range(args.steps + 2):
            a = time.perf_counter()
            it = native.CometExecIterator([native.DeviceInput(dt)], ncols, pb)
            b = time.perf_counter()
            out = []
            while True:
                r = native.Native.executePlan(it.handle, ncols)
                if r is None:
                    break
                out.append(r)
            c = time.perf_counter()
            st = it.kernel_stats()
            it.close()
            d = time.perf_counter()
            if i >= 2:
                tc += b - a
                te += c - b
                tr += d - c
                kms += st[0]
        n = args.steps
        k = kms / n
        print(f"{name}: rows={rows} gen={gen_s:.1f}s kernel={k:.3f} ms -> {rows / k / 1e6:.1f} Grows/s, algorithmic {rows * bpr / k / 1e6:.0f} GB/s "
              f"| host per step: create {tc / n * 1e3:.3f} ms, execute {te / n * 1e3:.3f} ms, release {tr / n * 1e3:.3f} ms", flush=True)
        del dt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
