#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_snappy_gpu.py tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
for codec in snappy; do
  COMET_TRACE_STAGES=1 timeout 600 python tools/parquet_q6.py --codec $codec --steps 5 --out $OUT/pq6_$codec.json > $OUT/pq6_$codec.log 2>&1
  cat $OUT/pq6_$codec.json
done
grep "\[comet\] parquet" $OUT/pq6_snappy.log | tail -12
