#!/bin/bash
# round 3: dictionary-encoded zstd pages on the device, index sections read back into the chunk's own pinned slot: SF10 Q6 from zstd Parquet
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $OUT /tmp/q6z
D=/tmp/q6z
export COMET_DEVICE_ZSTD_DICT=1
timeout 200 python -m pytest tests/test_device_zstd_gpu.py -x -q -k "dictionary or mixed" 2>&1 | tail -2 | cut -c1-200
for T in 16 1; do
  timeout 240 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads $T --steps 5 --out $OUT/q6_zstd_t${T}_dict1.json > $OUT/q6_zstd_t${T}_dict1.log 2>&1
  echo "threads=$T dict=1: $(cut -c1-260 $OUT/q6_zstd_t${T}_dict1.json)"
done
for T in 16 1; do
  COMET_TRACE_STAGES=1 timeout 120 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads $T --steps 2 > $OUT/q6_zstd_trace_t$T.log 2>&1
  echo "== threads $T"; grep "comet\] parquet" $OUT/q6_zstd_trace_t$T.log | tail -22 | cut -c1-150
done
COMET_DEVICE_ZSTD_DICT=0 COMET_TRACE_STAGES=1 timeout 120 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads 1 --steps 2 > $OUT/q6_zstd_trace_t1_dict0.log 2>&1
echo "== threads 1 dict 0"; grep "comet\] parquet" $OUT/q6_zstd_trace_t1_dict0.log | tail -12 | cut -c1-150
