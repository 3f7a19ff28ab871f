#!/bin/bash
# round 3, GPU call C: joins v3 (batched first-candidate checks), wave-aggregated group emit, pre-sized group tables
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hash_join_gpu.py tests/test_q95_gpu.py tests/test_final_agg_gpu.py tests/test_q1_gpu.py tests/test_tpch_more_gpu.py tests/test_q10_gpu.py tests/test_q36_gpu.py tests/test_fuzz_gpu.py tests/test_float_agg_gpu.py tests/test_expand_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --out $OUT/q3_fused.json > $OUT/q3_fused.log 2>&1
cat $OUT/q3_fused.json
timeout 300 python tools/q95_dist.py --orders 16000000 --steps 3 --warmup 1 --verify torch --out $OUT/q95.json > $OUT/q95.log 2>&1
cat $OUT/q95.json
cd /tmp
Q95="python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 2 --verify none"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats -o q95 -- $Q95 > $OUT/q95_stats.log 2>&1
head -9 $OUT/q95_stats/q95_kernel_stats.csv | cut -c1-100
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q95_fetch -o q95 -- $Q95 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/q95_tcc -o q95 -- $Q95 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/q95_sq -o q95 -- $Q95 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_join_summary.py $OUT > $OUT/q95_join_pmc.txt 2>&1
cd /tmp
Q3="python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- $Q3 > $OUT/q3_stats.log 2>&1
grep '^"k_' $OUT/q3_stats/q3_kernel_stats.csv | cut -c1-100
find $OUT -name "*.csv" -size +2M -delete
