"""The [bucket-INNER] case of tests/test_hash_join_gpu.py::test_fused_build_chain_matches_oracle_and_the_unfused_join, run K times, with the rows that differ
from the oracle's answer printed (which build rows, where they sit) — a diagnostic for a result that depended on the compiler (ROCm 7.2's lost two rows of 145 169)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datafusion_comet_amd  # noqa: F401
import numpy as np
import pyarrow as pa
from datafusion_comet_amd import native, serde as S
from oracle import oracle as O

CF = [S.T_INT64, S.T_INT32, S.T_INT64]
rng = np.random.default_rng(82)
nb, npr = 180_000, 90_000
keys = nb // 3
build_t = pa.table({"w": pa.array(rng.integers(-1000, 1000, nb), pa.int32(), mask=rng.random(nb) < 0.05), "k": pa.array(rng.integers(0, keys, nb), pa.int64(), mask=rng.random(nb) < 0.03),
                    "id": pa.array(np.arange(nb, dtype=np.int64))})
probe_t = pa.table({"k": pa.array(rng.integers(0, keys + keys // 4, npr), pa.int64(), mask=rng.random(npr) < 0.03), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()),
                    "id": pa.array(np.arange(npr, dtype=np.int64))})
BF = [S.T_INT32, S.T_INT64, S.T_INT64]
f1 = S.filter_(S.scan(BF), S.and_(S.gt(S.col(0, S.T_INT32), S.lit(-700, S.T_INT32)), S.is_not_null(S.col(1, S.T_INT64))))
pr = S.project(f1, [S.math("add", S.col(1, S.T_INT64), S.lit(0, S.T_INT64), S.T_INT64), S.col(2, S.T_INT64), S.math("multiply", S.col(0, S.T_INT32), S.lit(2, S.T_INT32), S.T_INT32)])
bchain = S.filter_(pr, S.lt(S.col(2, S.T_INT32), S.lit(1600, S.T_INT32)))
j = S.hash_join(S.scan(CF), bchain, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT, None)
want = O.run_plan_to_arrow(S, j, [probe_t, build_t])
wset = {}
for p, b in zip(want.column(2).to_pylist(), want.column(4).to_pylist()):
    wset[(p, b)] = wset.get((p, b), 0) + 1
print("toolchain:", native.jit_toolchain(), " want rows", want.num_rows, flush=True)
for mode in (sys.argv[1:] or ["always"]):
    cfg = S.config_map({"spark.comet.gpu.join.fuseBuild": mode})
    for rep in range(3):
        out = native.execute_to_table([native.HostInput.from_table(probe_t), native.HostInput.from_table(build_t)], 6, j.encode(), batch_size=0, config=cfg)
        got = pa.Table.from_batches(out)
        gset = {}
        for p, b in zip(got.column(2).to_pylist(), got.column(4).to_pylist()):
            gset[(p, b)] = gset.get((p, b), 0) + 1
        missing = [k for k in wset if gset.get(k, 0) < wset[k]]
        extra = [k for k in gset if wset.get(k, 0) < gset[k]]
        print(f"fuseBuild={mode} rep {rep}: rows {got.num_rows}  missing {len(missing)} extra {len(extra)}")
        bk, bw = build_t.column("k").to_pylist(), build_t.column("w").to_pylist()
        for p, b in missing[:6]:
            lo, hi = max(0, b - 3), min(nb, b + 4)
            print(f"   missing (probe id {p}, build id {b}) key {bk[b]}: build rows {lo}..{hi - 1}: k={bk[lo:hi]} w={bw[lo:hi]}  (row mod 64 = {b % 64})")
