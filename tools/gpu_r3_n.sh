#!/bin/bash
# batched uploads: parity, then SF10 Q6 from snappy Parquet with the upload kernel / one copy per slice / four copy streams, and the device timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_device_snappy_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python tools/parquet_q6.py --codec snappy --steps 10 --out $OUT/q6_$name.json > /dev/null 2> $OUT/q6_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/q6_$name.json'));print(round(d['sec_best']*1e3,2), round(d['sec_median']*1e3,2), [round(x*1e3,1) for x in d['sec_all']])")"
}
run kernel A=1
run copy1 COMET_PQ_UPLOAD=copy
run copy4 COMET_PQ_UPLOAD=copy COMET_PQ_COPY_STREAMS=4
run kernel_b A=1
run copy1_b COMET_PQ_UPLOAD=copy
run kernel_noinplace COMET_PARQUET_READ_IN_PLACE=0

COMET_TRACE_STAGES=1 timeout 300 python tools/parquet_q6.py --codec snappy --steps 2 > /dev/null 2> $OUT/trace.err
grep -E 'comet|tool' $OUT/trace.err | tail -14 | cut -c1-200
timeout 300 python tools/parquet_q6.py --codec zstd --steps 6 --out $OUT/q6_zstd.json > /dev/null 2>&1; cut -c1-300 $OUT/q6_zstd.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o pq -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec snappy --steps 2 > $OUT/pq.log 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT/tr/pq_kernel_trace.csv $OUT/tr/pq_memory_copy_trace.csv > $OUT/timeline.txt 2>&1
tail -60 $OUT/timeline.txt
find $OUT -name "*.csv" -size +3M -delete
