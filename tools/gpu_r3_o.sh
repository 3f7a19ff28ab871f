#!/bin/bash
# zstd on the device: parity (raw frames, scans), the pipeline alone on 480 pages of 1 MiB, SF10 Q6 from zstd Parquet device vs host, per-kernel times
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_zstd.log 2>&1
tail -15 $OUT/pytest_zstd.log | cut -c1-300
timeout 900 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_device_snappy_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
for lv in 1 3; do
  timeout 300 python tools/snappy_bench.py --codec zstd --level $lv --pages 480 --out $OUT/zstd_bench_l$lv.json > /dev/null 2> $OUT/zstd_bench_l$lv.err; cat $OUT/zstd_bench_l$lv.json; tail -2 $OUT/zstd_bench_l$lv.err
done
run() { name=$1; shift
  env "$@" timeout 300 python tools/parquet_q6.py --codec zstd --steps 8 --out $OUT/q6_$name.json > /dev/null 2> $OUT/q6_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/q6_$name.json'));print(round(d['sec_best']*1e3,2), round(d['sec_median']*1e3,2), [round(x*1e3,1) for x in d['sec_all']], d.get('match'))")"
}
run zstd_device COMET_DEVICE_DECOMPRESS=1
run zstd_host COMET_DEVICE_ZSTD=0
run zstd_auto A=1
run zstd_device_1thread COMET_DEVICE_DECOMPRESS=1 COMET_SCAN_THREADS=1
run zstd_auto_1thread COMET_SCAN_THREADS=1
run zstd_host_1thread COMET_DEVICE_ZSTD=0 COMET_SCAN_THREADS=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zs -o zs -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --codec zstd --level 1 --pages 480 --no-check > /dev/null 2>&1
grep -E 'zs2_' $OUT/zs/zs_kernel_stats.csv | cut -c1-160
timeout 600 COMET_DEVICE_DECOMPRESS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o pq -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec zstd --steps 2 > $OUT/pq.log 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT/tr/pq_kernel_trace.csv $OUT/tr/pq_memory_copy_trace.csv > $OUT/timeline.txt 2>&1
tail -50 $OUT/timeline.txt
find $OUT -name "*.csv" -size +3M -delete
