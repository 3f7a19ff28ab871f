#!/usr/bin/env python3
"""Where does the Q1 kernel's time go?  Runs Q1 variants (string keys vs int keys, with/without the 256-bit
multiply, fewer aggregates) on the same resident data and prints kernel times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pyarrow as pa


def main():
    from datafusion_comet_amd import native, serde as S, tpch
    rows = int(os.environ.get("ROWS", "50000000"))
    t = tpch.lineitem_q1(rows)
    rf = np.frombuffer(t.column("l_returnflag").chunk(0).buffers()[2], np.uint8)[:rows].astype(np.int32)
    ls = np.frombuffer(t.column("l_linestatus").chunk(0).buffers()[2], np.uint8)[:rows].astype(np.int32)
    t = t.append_column("rf_i", pa.array(rf)).append_column("ls_i", pa.array(ls))
    dt = native.DeviceTable.from_arrow(t, "cuda:0")
    DEC = tpch.DEC
    fields = [DEC, DEC, DEC, DEC, S.T_STRING, S.T_STRING, S.T_DATE, S.T_INT32, S.T_INT32]
    qty, price, disc, tax, rfs, lss, ship, rfi, lsi = (S.col(i, f) for i, f in enumerate(fields))

    def plan(str_keys=True, wide=True, nagg=8, filt=True):
        src = S.scan(fields)
        if filt:
            src = S.filter_(src, S.lt_eq(ship, S.lit(tpch.days(1998, 9, 2), S.T_DATE)))
        one = S.lit(100, DEC)
        om = S.check_overflow(S.math("subtract", one, disc, S.decimal(13, 2)), S.decimal(13, 2))
        op = S.check_overflow(S.math("add", one, tax, S.decimal(13, 2)), S.decimal(13, 2))
        dp = S.check_overflow(S.math("multiply", price, om, S.decimal(26, 4)), S.decimal(26, 4))
        ch = S.check_overflow(S.math("multiply", dp, op, S.decimal(38, 6)), S.decimal(38, 6))
        keys = [rfs, lss] if str_keys else [rfi, lsi]
        kt = [S.T_STRING] * 2 if str_keys else [S.T_INT32] * 2
        p = S.project(src, keys + [qty, price, disc, dp] + ([ch] if wide else []))
        D22, D16 = S.decimal(22, 2), S.decimal(16, 6)
        c = lambda i, ty: S.col(i, ty)
        aggs = [S.sum_(c(2, DEC), D22), S.sum_(c(3, DEC), D22), S.sum_(c(5, S.decimal(26, 4)), S.decimal(36, 4))]
        if wide:
            aggs.append(S.sum_(c(6, S.decimal(38, 6)), S.decimal(38, 6)))
        aggs += [S.avg(c(2, DEC), D16, D22), S.avg(c(3, DEC), D16, D22), S.avg(c(4, DEC), D16, D22), S.count(S.lit(1, S.T_INT32))]
        aggs = aggs[:nagg]
        ncols = 2 + sum(2 if a.kind in ("sum", "avg") else 1 for a in aggs)
        return S.hash_agg(p, [c(0, kt[0]), c(1, kt[1])], aggs), ncols

    for name, kw in (("full", {}), ("int keys", dict(str_keys=False)), ("no wide mul", dict(wide=False)), ("int keys, no wide", dict(str_keys=False, wide=False)),
                     ("int keys, no wide, 1 agg", dict(str_keys=False, wide=False, nagg=1)), ("full, no filter", dict(filt=False))):
        pl, ncols = plan(**kw)
        pb = pl.encode()
        ms = []
        for i in range(6):
            it = native.CometExecIterator([native.DeviceInput(dt)], ncols, pb)
            while native.Native.executePlan(it.handle, ncols) is not None:
                pass
            ms.append(it.kernel_stats()[0])
            it.close()
        k = min(ms[1:])
        print(f"{name:28s} kernel {k:.3f} ms  {rows / k / 1e6:.1f} Grows/s", flush=True)


if __name__ == "__main__":
    main()
