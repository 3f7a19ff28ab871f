"""TPC-DS Q95 stage A on one GPU, run by run: stage time, the join kernels' event times and the plan's join metrics (which table each join took).
A diagnostic: python tools/q95_metrics.py [--orders N] [--runs K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--orders", type=int, default=16_000_000)
ap.add_argument("--runs", type=int, default=4)
a = ap.parse_args()
import pyarrow as pa
import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
import torch
from datafusion_comet_amd import native, tpcds
from datafusion_comet_amd import serde as S

cache = f"/tmp/q95_tables_{a.orders}"
if os.path.isdir(cache):
    t = {f[:-6]: pa.ipc.open_file(pa.memory_map(os.path.join(cache, f))).read_all() for f in os.listdir(cache) if f.endswith(".arrow")}
else:
    t = tpcds.q95_tables(a.orders)
stage_a, _, leaves = tpcds.q95_plans()
dev = {k: native.DeviceTable.from_arrow(v) for k, v in t.items()}
plan = stage_a.encode()


def flat(node, acc):
    m, kids = node
    for k, v in m.items():
        if k.startswith("join_") or k.startswith("agg_part"):
            acc[k] = acc.get(k, 0) + v
    for c in kids:
        flat(c, acc)
    return acc


for r in range(a.runs):
    inputs = [native.DeviceInput(dev[n]) for n in leaves]
    torch.cuda.synchronize()
    with native.collect_kernel_times() as kt:
        t0 = time.perf_counter()
        it = native.CometExecIterator(inputs, 5, plan, batch_size=0)
        rows = 0
        while True:
            b = native.Native.executePlan(it.handle, 5)
            if b is None:
                break
            rows += b.num_rows
        ms = (time.perf_counter() - t0) * 1e3
        met = flat(S.decode_metric_node(it.metrics()), {})
        it.close()
    for i in inputs:
        i.close()
    ks = {k: (round(v["ms"], 3), v["calls"]) for k, v in sorted(kt.times.items(), key=lambda e: -e[1]["ms"])[:6]}
    print(json.dumps({"run": r, "stage_a_ms_with_event_pairs": round(ms, 2), "kernels": ks, "metrics": met}), flush=True)
