"""TPC-DS Q95 stage A + B on one GPU with the library and Python package of ANOTHER checkout (--root: a directory holding datafusion-comet_amd/ with a
built libcomet.so and datafusion_comet_amd/): the bisect vehicle of the round-4 Q95 regression.  Same loop as tools/q95_bench.py; prints one JSON line."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--root", required=True)
ap.add_argument("--orders", type=int, default=16_000_000)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--tag", default="")
a = ap.parse_args()
sys.path.insert(0, os.path.abspath(a.root))
import pyarrow as pa
import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
import torch
from datafusion_comet_amd import native, tpcds
cache = f"/tmp/q95_tables_{a.orders}"
if os.path.isdir(cache):      # generated once per box, memory-mapped by the later variants (generation is 30 s of page faults)
    t = {f[:-6]: pa.ipc.open_file(pa.memory_map(os.path.join(cache, f))).read_all() for f in os.listdir(cache) if f.endswith(".arrow")}
else:
    t = tpcds.q95_tables(a.orders)
    os.makedirs(cache + ".tmp", exist_ok=True)
    for k, v in t.items():
        with pa.OSFile(os.path.join(cache + ".tmp", k + ".arrow"), "wb") as f, pa.ipc.new_file(f, v.schema) as w:
            w.write_table(v)
    os.rename(cache + ".tmp", cache)
stage_a, stage_b, leaves = tpcds.q95_plans()
dev = {k: native.DeviceTable.from_arrow(v) for k, v in t.items()}
pa_bytes, pb_bytes = stage_a.encode(), stage_b.encode()
times, res = [], None
for r in range(a.reps + 1):
    inputs = [native.DeviceInput(dev[n]) for n in leaves]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = pa.Table.from_batches(native.execute_to_table(inputs, 5, pa_bytes, batch_size=0))
    t1 = time.perf_counter()
    res = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(st)], 3, pb_bytes, batch_size=0))
    t2 = time.perf_counter()
    for i in inputs:
        i.close()
    if r > 0:
        times.append((t2 - t0, t1 - t0))
print(json.dumps({"tag": a.tag or a.root, "lib": native.lib()._name if hasattr(native.lib(), "_name") else "", "stage_a_ms": [round(x[1] * 1e3, 2) for x in times],
                  "total_ms_best": round(min(x[0] for x in times) * 1e3, 2), "result": [str(res.column(i)[0].as_py()) for i in range(3)]}))
