"""Timings of the operators around the hot path on one GPU with HBM-resident inputs: group-by on long Utf8 keys, Window (rank + running decimal
sum) and ORDER BY a Utf8 key.  Usage: python tools/micro_bench.py [--rows N] [--out json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50_000_000)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import numpy as np
    import pyarrow as pa
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    import __graft_entry__ as g
    g.build()
    from datafusion_comet_amd import native, serde as S
    n, dev = a.rows, "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(11)
    # 24-byte names "Customer#%015d" built on the device: ids → decimal digits
    nd = 1_000_000
    ids = torch.randint(0, nd, (n,), generator=gen, device=dev, dtype=torch.int64)
    digits = torch.stack([(ids // (10 ** k)) % 10 for k in range(14, -1, -1)], dim=1).to(torch.uint8) + 48
    prefix = torch.tensor(list(b"Customer#"), dtype=torch.uint8, device=dev).repeat(n, 1)
    data = torch.cat([prefix, digits], dim=1).contiguous().reshape(-1)
    offs = (torch.arange(n + 1, device=dev, dtype=torch.int64) * 24).to(torch.int32)
    amount = torch.randint(-10**9, 10**9, (n,), generator=gen, device=dev, dtype=torch.int64)
    dec = torch.stack([amount, amount >> 63], dim=1).contiguous().view(torch.uint8).reshape(-1)
    part = torch.sort(torch.randint(0, 100_000, (n,), generator=gen, device=dev, dtype=torch.int32)).values     # window input: sorted by partition key
    schema = pa.schema([("name", pa.string()), ("v", pa.decimal128(12, 2)), ("p", pa.int32())])
    table = native.DeviceTable(schema, n, [offs.view(torch.uint8), dec, part.view(torch.uint8)], [None] * 3, dev, aux=[data, None, None])
    D, SD = S.decimal(12, 2), S.decimal(22, 2)
    fields = [S.T_STRING, D, S.T_INT32]
    name, v, p = (S.col(i, t) for i, t in enumerate(fields))
    res = {"rows": n}

    def timed(label, plan, ncols, nbytes):
        best = None
        for r in range(a.reps + 1):
            inp = native.DeviceInput(table)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = native.execute_to_device([inp], ncols, plan.encode())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rows = out.num_rows
            del out
            inp.close()
            if r:
                best = dt if best is None else min(best, dt)
        res[label] = {"seconds": best, "rows_per_s": n / best, "GB_per_s": nbytes / best / 1e9, "out_rows": rows}
        print(label, res[label], flush=True)

    timed("group_by_24_byte_strings_1M_groups", S.hash_agg(S.scan(fields), [name], [S.sum_(v, SD), S.count(v)], S.PARTIAL), 4, n * (4 + 24 + 16))
    timed("window_rank_and_running_sum", S.window(S.scan(fields), [p], [(v, True, True)],
                                                  [("rank", [], S.T_INT32), ("agg", S.sum_(v, SD), SD, ("range", "unbounded", "current"))]), 5, n * (16 + 4 + 4 + 16))
    n2 = min(n, 20_000_000)
    small = native.DeviceTable(schema, n2, [offs[:n2 + 1].contiguous().view(torch.uint8), dec[:n2 * 16].contiguous(), part[:n2].contiguous().view(torch.uint8)], [None] * 3, dev,
                               aux=[data[:n2 * 24].contiguous(), None, None])
    table = small
    n_save, n = n, n2
    timed("order_by_24_byte_string_top100", S.sort(S.scan(fields), [(name, False, False), (v, True, True)], fetch=100), 3, n2 * (4 + 24 + 16))
    timed("order_by_24_byte_string_full", S.sort(S.scan(fields), [(name, False, False)]), 3, n2 * (4 + 24 + 16 + 4) * 2)
    res["sort_rows"] = n2
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
