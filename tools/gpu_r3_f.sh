#!/bin/bash
# round 3, GPU call F: where the fragment kernel's time goes (phases switched off one at a time; outputs unchecked)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd /tmp
for skip in 0 1 2 3 8 4; do
  COMET_SN2_DEBUG_SKIP=$skip timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$skip -o sn -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --pages 480 --skip-one-wave --no-check --kinds decimal_int64 > /dev/null 2>&1
  echo "skip=$skip: $(grep sn2_exec $OUT/s$skip/sn_kernel_stats.csv | cut -d, -f5-7)"
done
find $OUT -name "*.csv" -size +1M -delete
