#!/bin/bash
# round 3, GPU call H: where SF10 Q6 from Parquet spends its time (stage trace), snappy / zstd / none
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $OUT
for codec in snappy zstd none; do
  COMET_TRACE_STAGES=1 timeout 600 python tools/parquet_q6.py --codec $codec --steps 3 --out $OUT/pq6_$codec.json > $OUT/pq6_$codec.log 2>&1
  cat $OUT/pq6_$codec.json
done
grep "\[comet\]" $OUT/pq6_snappy.log | tail -40
