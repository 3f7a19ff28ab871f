#!/bin/bash
# checkpoint: the whole GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3full
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log | cut -c1-300
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3full/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","ms_per_step","vs_baseline")})
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic")})
print("q6_sf10_parquet", json.dumps(d.get("q6_sf10_parquet"))[:900])
print("snappy", json.dumps(d.get("snappy_pipeline"))[:400])
print("zstd", json.dumps(d.get("zstd_pipeline"))[:400])
print("q3", json.dumps(d.get("q3"))[:300])
print("q95", json.dumps(d.get("q95"))[:300])
print("paths", json.dumps(d.get("paths"))[:500])
PY
