#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $OUT
run() { name=$1; shift
  env "$@" timeout 300 python tools/parquet_q6.py --codec snappy --steps 10 --out $OUT/q6_$name.json > /dev/null 2> $OUT/q6_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/q6_$name.json'));print(round(d['sec_best']*1e3,2), round(d['sec_median']*1e3,2), [round(x*1e3,1) for x in d['sec_all']])")"
}
run big_first A=1
run small_first COMET_PQ_ORDER=small_first
run big_first_b A=1
run small_first_b COMET_PQ_ORDER=small_first
COMET_PQ_ORDER=small_first COMET_TRACE_STAGES=1 timeout 300 python tools/parquet_q6.py --codec snappy --steps 2 > /dev/null 2> $OUT/trace.err
grep -E 'comet' $OUT/trace.err | tail -12 | cut -c1-200
