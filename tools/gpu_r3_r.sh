#!/bin/bash
# page-pruned scans through the run-at-a-time decode kernel (clipped units): parity, then a date-clustered SF10 Q6 file with the date range pushed down
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet_page_index_gpu.py tests/test_parquet_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_parquet_fixtures_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
COMET_PQ_DECODE_ROWS=1 timeout 600 python -m pytest tests/test_parquet_page_index_gpu.py -x -q > $OUT/pytest_rows.log 2>&1
tail -2 $OUT/pytest_rows.log | cut -c1-300
for codec in none snappy; do
  for mode in units rows; do
    if [ $mode = rows ]; then export COMET_PQ_DECODE_ROWS=1; else unset COMET_PQ_DECODE_ROWS; fi
    timeout 300 python tools/parquet_q6.py --codec $codec --clustered --steps 6 --out $OUT/q6_clustered_${codec}_$mode.json > /dev/null 2> $OUT/q6_clustered_${codec}_$mode.err
    echo "$codec $mode: $(python -c "import json;d=json.load(open('$OUT/q6_clustered_${codec}_$mode.json'));print(round(d['sec_best']*1e3,2), round(d['sec_median']*1e3,2), d.get('result'))")"
  done
done
unset COMET_PQ_DECODE_ROWS
cd /tmp
for mode in units rows; do
  if [ $mode = rows ]; then export COMET_PQ_DECODE_ROWS=1; else unset COMET_PQ_DECODE_ROWS; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$mode -o k -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec none --clustered --steps 4 > /dev/null 2>&1
  echo "== $mode"; grep -E 'pq_decode|pq_validity|pq_vidx|pq_expand' $OUT/k_$mode/k_kernel_stats.csv | cut -c1-150
done
find $OUT -name "*.csv" -size +3M -delete
