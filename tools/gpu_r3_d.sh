#!/bin/bash
# round 3, GPU call D: the multi-kernel snappy pipeline (tests, kernel-alone bench, SF10 Q6 from Parquet), host-stream ramp, PCIe probe, Q3 trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_snappy_gpu.py tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py -x -q -s > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/snappy_bench.py --pages 480 --out $OUT/snappy_bench.json > $OUT/snappy_bench.log 2>&1
cat $OUT/snappy_bench.json
timeout 300 python tools/h2d_probe.py > $OUT/h2d.json 2> $OUT/h2d.err
cat $OUT/h2d.json
COMET_DEVICE_DECOMPRESS=1 timeout 600 python tools/parquet_q6.py --codec snappy --out $OUT/pq6_snappy.json > $OUT/pq6_snappy.log 2>&1
cat $OUT/pq6_snappy.json
COMET_DEVICE_DECOMPRESS=1 COMET_SNAPPY_ONE_WAVE=1 timeout 600 python tools/parquet_q6.py --codec snappy --out $OUT/pq6_snappy_onewave.json > $OUT/pq6_snappy_onewave.log 2>&1
cat $OUT/pq6_snappy_onewave.json
COMET_DEVICE_DECOMPRESS=0 timeout 600 python tools/parquet_q6.py --codec snappy --out $OUT/pq6_snappy_host.json > $OUT/pq6_snappy_host.log 2>&1
cat $OUT/pq6_snappy_host.json
timeout 600 python tools/paths.py --query q1 --rows 20000000 --out $OUT/paths.json > $OUT/paths.log 2>&1
cat $OUT/paths.json
COMET_TRACE_STAGES=1 timeout 300 python tools/q3_dist.py --orders 150000000 --steps 1 --warmup 1 --no-verify 2>&1 | grep "grouped result\|materialize" | tail -8
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn_stats -o sn -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --pages 480 --skip-one-wave > /dev/null 2>&1
grep "sn2_\|pq_snappy" $OUT/sn_stats/sn_kernel_stats.csv | cut -c1-160
find $OUT -name "*.csv" -size +2M -delete
