#!/bin/bash
# round 3, final measurement run: the bench line, its rocprof view, per-query kernel statistics, PMC of the join kernels, snappy / PCIe probes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $OUT
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $OUT/bench_stats.json 2> /dev/null
grep -E '^"k_|utf8_uniform' $OUT/bench_stats/b_kernel_stats.csv | cut -c1-110
Q3="python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- $Q3 > $OUT/q3_stats.log 2>&1
grep '^"k_' $OUT/q3_stats/q3_kernel_stats.csv | cut -c1-110
Q95="python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 2 --verify none"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats -o q95 -- $Q95 > $OUT/q95_stats.log 2>&1
grep '^"k_' $OUT/q95_stats/q95_kernel_stats.csv | cut -c1-110
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q95_fetch -o q95 -- $Q95 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/q95_tcc -o q95 -- $Q95 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/q95_sq -o q95 -- $Q95 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_join_summary.py $OUT > $OUT/q95_join_pmc.txt 2>&1
head -30 $OUT/q95_join_pmc.txt | cut -c1-250
timeout 300 python tools/snappy_bench.py --pages 480 --out $OUT/snappy_bench.json > /dev/null 2>&1; cat $OUT/snappy_bench.json
timeout 300 python tools/h2d_probe.py > $OUT/h2d.json 2>/dev/null; cat $OUT/h2d.json
timeout 300 python tools/filter_bench.py > $OUT/filter_bench.json 2>/dev/null; cat $OUT/filter_bench.json
find $OUT -name "*.csv" -size +2M -delete
