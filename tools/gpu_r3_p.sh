#!/bin/bash
# zstd pipeline: per-kernel times on 480 pages of 1 MiB (decimal only), parity of the raw-frame tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_zstd.log 2>&1
tail -3 $OUT/pytest_zstd.log | cut -c1-300
timeout 300 python tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --out $OUT/zstd_bench_l1.json > /dev/null 2> $OUT/zstd_bench_l1.err; cat $OUT/zstd_bench_l1.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zs -o zs -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --codec zstd --level 1 --pages 480 --no-check --kinds decimal_int64 > /dev/null 2>&1
grep -E 'zs2_' $OUT/zs/zs_kernel_stats.csv | sed 's/(comet_zstd2[^"]*"/"/' | cut -c1-160
find $OUT -name "*.csv" -size +3M -delete
