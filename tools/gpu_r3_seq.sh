#!/bin/bash
# round 3, last GPU seconds: the lean sequence step (bit cursor + three-word windows) — parity on the device, then the 480-page pipeline timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3seq
mkdir -p $OUT
timeout 60 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_zstd.log 2>&1
tail -2 $OUT/pytest_zstd.log | cut -c1-200
timeout 40 python tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --out $OUT/zstd_bench_l1.json > /dev/null 2> $OUT/zstd_bench_l1.err; cat $OUT/zstd_bench_l1.json
