#!/bin/bash
# round 3: where one scan thread's time goes on SF10 Q6 from zstd / snappy Parquet (COMET_TRACE_STAGES host timers)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z2
mkdir -p $OUT /tmp/q6z
D=/tmp/q6z
for CFG in "zstd 1 1" "zstd 1 0" "zstd 16 0" "snappy 1 0"; do
  set -- $CFG
  COMET_DEVICE_ZSTD_DICT=$3 COMET_TRACE_STAGES=1 timeout 120 python tools/parquet_q6.py --codec $1 --dir $D --scan-threads $2 --steps 2 > $OUT/trace_$1_t$2_dict$3.log 2>&1
  echo "== $CFG"; grep "scan threads spent\|all launches\|decompressed on" $OUT/trace_$1_t$2_dict$3.log | tail -3 | cut -c1-200; grep -o '"file_bytes": [0-9]*' $OUT/trace_$1_t$2_dict$3.log | tail -1
done
