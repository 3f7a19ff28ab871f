#!/bin/bash
# in-place chunk reads: parity of the Parquet / snappy suites, then SF10 Q6 from snappy Parquet with and without
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_device_snappy_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
for ip in 1 0 1 0; do
  COMET_PARQUET_READ_IN_PLACE=$ip timeout 300 python tools/parquet_q6.py --codec snappy --steps 8 --out $OUT/q6_snappy_ip$ip.json > /dev/null 2> $OUT/q6_snappy_ip$ip.err
  echo "in_place=$ip"; cut -c1-400 $OUT/q6_snappy_ip$ip.json
done
COMET_TRACE_STAGES=1 COMET_TRACE=1 timeout 300 python tools/parquet_q6.py --codec snappy --steps 2 > /dev/null 2> $OUT/trace.err
grep -E 'comet|tool' $OUT/trace.err | tail -40 | cut -c1-200
