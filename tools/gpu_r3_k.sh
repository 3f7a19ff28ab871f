#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_q95_gpu.py tests/test_hash_join_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/q95_dist.py --orders 16000000 --steps 3 --warmup 1 --verify torch --out $OUT/q95.json > $OUT/q95.log 2>&1
cat $OUT/q95.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dec -o dec -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec zstd --steps 3 > $OUT/dec.log 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_trace_tail.py $OUT/dec/dec_kernel_trace.csv pq_decode pq_expand
find $OUT -name "*.csv" -size +2M -delete
