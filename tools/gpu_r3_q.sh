#!/bin/bash
# rows per thread of the fused filter kernel (COMET_EXPERIMENT=R) on the Q3 stages; what HBM gives plain fill / copy / sum kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $OUT
timeout 200 python tools/hbm_probe.py $OUT/hbm_probe.json
for R in 8 12 16 4; do
  echo "R=$R"; COMET_EXPERIMENT=$R timeout 400 python tools/filter_bench.py --reps 3 --out $OUT/filter_R$R.json 2>&1 | grep -E '^(lineitem|orders|config1)' | cut -c1-260
done
