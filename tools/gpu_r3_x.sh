#!/bin/bash
# round 3: dictionary-encoded zstd pages on the device — parity first, then SF10 Q6 from zstd Parquet with and without it, then the whole GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3x
mkdir -p $OUT
timeout 300 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_zstd.log 2>&1
tail -3 $OUT/pytest_zstd.log | cut -c1-300
if ! grep -q ' passed' $OUT/pytest_zstd.log || grep -q 'failed\|error' $OUT/pytest_zstd.log; then tail -40 $OUT/pytest_zstd.log | cut -c1-250; exit 1; fi
D=/tmp/q6z
for T in 16 1; do
  for DICT in 1 0; do
    COMET_DEVICE_ZSTD_DICT=$DICT timeout 240 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads $T --steps 5 --out $OUT/q6_zstd_t${T}_dict${DICT}.json > $OUT/q6_zstd_t${T}_dict${DICT}.log 2>&1
    echo "threads=$T dict=$DICT: $(cut -c1-420 $OUT/q6_zstd_t${T}_dict${DICT}.json)"
  done
done
COMET_TRACE_STAGES=1 timeout 120 python tools/parquet_q6.py --codec zstd --dir $D --scan-threads 1 --steps 1 > $OUT/q6_zstd_trace.log 2>&1
grep -c "index sections" $OUT/q6_zstd_trace.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
grep -E 'passed|failed|error' $OUT/pytest_gpu.log | tail -3 | cut -c1-200
