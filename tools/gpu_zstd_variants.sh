#!/bin/bash
# On the GPU box (gpurun -- 'bash tools/gpu_zstd_variants.sh'): the shipped library, then each variant built by tools/build_zstd_variants.sh
# in its place (the box works on a scratch copy of the tree): device parity (tests/test_device_zstd_gpu.py) and the 480-page pipeline timing.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/zstd_variants
mkdir -p $OUT
cp datafusion-comet_amd/libcomet.so /tmp/libcomet_shipped.so
for v in shipped $(ls datafusion-comet_amd/variants 2>/dev/null | sed -n 's/^libcomet_\(.*\)\.so$/\1/p'); do
  if [ $v = shipped ]; then cp /tmp/libcomet_shipped.so datafusion-comet_amd/libcomet.so; else cp datafusion-comet_amd/variants/libcomet_$v.so datafusion-comet_amd/libcomet.so; fi
  timeout 120 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_$v.log 2>&1
  echo "== $v: $(tail -1 $OUT/pytest_$v.log | cut -c1-80)"
  for rep in 1 2; do
    timeout 60 python tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --out $OUT/bench_${v}_$rep.json > /dev/null 2> $OUT/bench_${v}_$rep.err
    cut -c1-260 $OUT/bench_${v}_$rep.json; echo
  done
done
cp /tmp/libcomet_shipped.so datafusion-comet_amd/libcomet.so
