#!/bin/bash
# round 3, closing run: the whole GPU suite, the default bench line, its rocprof view, per-query kernel statistics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final2
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
grep -E 'passed|failed|error' $OUT/pytest_gpu.log | tail -3 | cut -c1-200
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $OUT/bench_stats.json 2> /dev/null
grep -E '^"k_|utf8_uniform' $OUT/bench_stats/b_kernel_stats.csv | cut -c1-110
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify > $OUT/q3_stats.log 2>&1
grep '^"k_' $OUT/q3_stats/q3_kernel_stats.csv | cut -c1-110
cd $GRAFT_REPO_ROOT
timeout 300 python tools/snappy_bench.py --pages 480 --out $OUT/snappy_bench.json > /dev/null 2>&1; cut -c1-600 $OUT/snappy_bench.json
find $OUT -name "*.csv" -size +2M -delete
