#!/bin/bash
# round 3: SF10 Q6 from zstd Parquet, dictionary-encoded pages on the device vs on host threads, 16 scan threads and 1; then the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $OUT /tmp/q6z
D=/tmp/q6z
for T in 16 1; do
  for DICT in 1 0; do
    COMET_DEVICE_ZSTD_DICT=$DICT timeout 240 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads $T --steps 5 --out $OUT/q6_zstd_t${T}_dict${DICT}.json > $OUT/q6_zstd_t${T}_dict${DICT}.log 2>&1
    echo "threads=$T dict=$DICT: $(cut -c1-420 $OUT/q6_zstd_t${T}_dict${DICT}.json)"
  done
done
COMET_TRACE_STAGES=1 timeout 120 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads 1 --steps 1 > $OUT/q6_zstd_trace.log 2>&1
grep "index sections\|decompressed on" $OUT/q6_zstd_trace.log | sort | uniq -c | head
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
