#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o pq -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec snappy --steps 2 > $OUT/pq.log 2>&1
ls -la $OUT/tr
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT/tr/pq_kernel_trace.csv $OUT/tr/pq_memory_copy_trace.csv > $OUT/timeline.txt 2>&1
tail -80 $OUT/timeline.txt
find $OUT -name "*.csv" -size +3M -delete
