#!/bin/bash
# lanes (blocks) per wave of the zstd sequence kernel: 2 / 4 / 8
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3x
mkdir -p $OUT
cp datafusion-comet_amd/libcomet.so /tmp/libcomet_l4.so
for N in 2 4 8; do
  if [ $N != 4 ]; then cp datafusion-comet_amd/libcomet_l$N.so datafusion-comet_amd/libcomet.so; else cp /tmp/libcomet_l4.so datafusion-comet_amd/libcomet.so; fi
  timeout 300 python tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --out $OUT/l$N.json > /dev/null 2> $OUT/l$N.err
  echo "lanes=$N $(python -c "import json;d=json.load(open('$OUT/l$N.json'));print(d['decimal_int64']['pipeline'])")"
done
cp /tmp/libcomet_l4.so datafusion-comet_amd/libcomet.so
