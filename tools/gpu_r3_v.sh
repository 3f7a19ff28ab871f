#!/bin/bash
cd /tmp
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec none --steps 4"
timeout 300 $CMD > /dev/null 2>&1
for mul in 2048 1024 512; do
  COMET_PQ_PLAIN_CHUNK=$mul timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$mul -o k -- $CMD > /dev/null 2>&1
  echo "plain_chunk=$mul"; python $GRAFT_REPO_ROOT/tools/kernel_trace_tail.py $OUT/k$mul/k_kernel_trace.csv pq_decode | tail -6
done
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py -x -q 2>&1 | tail -2
find $OUT -name "*.csv" -size +3M -delete
