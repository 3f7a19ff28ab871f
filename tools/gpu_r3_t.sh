#!/bin/bash
# dictionary pages inflated on the device; scanThreads bounding the workers: parity, SF10 Q6 from snappy / zstd Parquet
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_page_index_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_device_snappy_gpu.py tests/test_device_zstd_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log | cut -c1-300
run() { name=$1; shift
  timeout 300 python tools/parquet_q6.py "$@" --out $OUT/q6_$name.json > /dev/null 2> $OUT/q6_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/q6_$name.json'));print(round(d['sec_best']*1e3,2), round(d['sec_median']*1e3,2), [round(x*1e3,1) for x in d['sec_all']], d['pages_decompressed_on_device'], d['matches_resident_plan'])")"
}
run snappy --codec snappy --steps 10
run snappy_b --codec snappy --steps 10
run zstd_auto --codec zstd --steps 6
run zstd_task_device --codec zstd --steps 4 --scan-threads 1
run zstd_task_host --codec zstd --steps 4 --scan-threads 1 --device-decompress false
run snappy_task_device --codec snappy --steps 4 --scan-threads 1
run snappy_task_host --codec snappy --steps 4 --scan-threads 1 --device-decompress false
