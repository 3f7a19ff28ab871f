#!/bin/bash
# round 3, GPU call B: the rewritten join kernels (runs, compaction, batched heads), semi-join reduction, TCP exchange; Q3 / Q95 numbers
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hash_join_gpu.py tests/test_q95_gpu.py tests/test_native_exchange_gpu.py tests/test_tpch_more_gpu.py tests/test_q10_gpu.py tests/test_q36_gpu.py tests/test_fuzz_gpu.py tests/test_utf8_passthrough_gpu.py tests/test_exchange_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 300 python tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --out $OUT/q3_fused.json > $OUT/q3_fused.log 2>&1
timeout 300 python tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --plan staged --out $OUT/q3_staged.json > $OUT/q3_staged.log 2>&1
COMET_TRACE_STAGES=1 timeout 300 python tools/q3_dist.py --orders 150000000 --steps 1 --warmup 1 --no-verify > $OUT/q3_fused_trace.log 2>&1
cat $OUT/q3_fused.json $OUT/q3_staged.json
tail -22 $OUT/q3_fused_trace.log | grep materialize
timeout 300 python tools/q95_dist.py --orders 16000000 --steps 3 --warmup 1 --verify torch --out $OUT/q95.json > $OUT/q95.log 2>&1
cat $OUT/q95.json
COMET_JOIN_SEMI_REDUCTION=0 timeout 300 python tools/q95_dist.py --orders 16000000 --steps 3 --warmup 1 --verify torch --out $OUT/q95_noreduce.json > $OUT/q95_noreduce.log 2>&1
cat $OUT/q95_noreduce.json
cd /tmp
Q95="python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 2 --verify none"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats -o q95 -- $Q95 > $OUT/q95_stats.log 2>&1
head -12 $OUT/q95_stats/q95_kernel_stats.csv
Q3="python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- $Q3 > $OUT/q3_stats.log 2>&1
head -14 $OUT/q3_stats/q3_kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete
