#!/bin/bash
# round 3, GPU call E: snappy pipeline after the first optimisation pass
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_snappy_gpu.py -x -q -s > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/snappy_bench.py --pages 480 --out $OUT/snappy_bench.json > $OUT/snappy_bench.log 2>&1
cat $OUT/snappy_bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn_stats -o sn -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --pages 480 --skip-one-wave > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_trace_tail.py $OUT/sn_stats/sn_kernel_trace.csv sn2_ snappy
cd $GRAFT_REPO_ROOT
COMET_DEVICE_DECOMPRESS=1 timeout 600 python tools/parquet_q6.py --codec snappy --out $OUT/pq6_snappy.json > $OUT/pq6_snappy.log 2>&1
cat $OUT/pq6_snappy.json
find $OUT -name "*.csv" -size +2M -delete
