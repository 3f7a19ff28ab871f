#!/bin/bash
# round 3, GPU call G: host-stream staging with streaming stores, block-level group emit, snappy two-hop rounds; Q3 / Q95 / paths; tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_snappy_gpu.py tests/test_q1_gpu.py tests/test_final_agg_gpu.py tests/test_scan_cast_gpu.py tests/test_aligned_import_gpu.py tests/test_dictionary_input_gpu.py tests/test_utf8_passthrough_gpu.py tests/test_q3_gpu.py tests/test_hash_join_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/paths.py --query q1 --rows 20000000 --out $OUT/paths.json > $OUT/paths.log 2>&1
cat $OUT/paths.json
timeout 300 python tools/host_path.py > $OUT/host_path.log 2>&1; tail -3 $OUT/host_path.log
timeout 300 python tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --out $OUT/q3_fused.json > $OUT/q3_fused.log 2>&1
cat $OUT/q3_fused.json
timeout 600 python tools/snappy_bench.py --pages 480 --skip-one-wave --out $OUT/snappy_bench.json > $OUT/snappy_bench.log 2>&1
cat $OUT/snappy_bench.json
cd /tmp
Q3="python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- $Q3 > $OUT/q3_stats.log 2>&1
grep '^"k_' $OUT/q3_stats/q3_kernel_stats.csv | cut -c1-100
find $OUT -name "*.csv" -size +2M -delete
