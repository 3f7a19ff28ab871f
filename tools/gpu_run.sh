#!/bin/bash
# The one GPU-box script: gpurun -- 'bash tools/gpu_run.sh <tag> <section> [<section> …]'.  Each section is a function below; its output goes
# under gpurun_out/<tag>/ (merged back by gpurun), a short tail to stdout.  Sections keep their own timeouts so a hung kernel costs one section.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
PQ=/tmp/q6pq; mkdir -p $PQ

tests() {          # the whole GPU suite
  timeout ${TESTS_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --timeout ${TEST_TIMEOUT:-300} --timeout-method=thread > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log | cut -c1-200
}
tests_sel() {      # tests matched by $SEL (a -k expression) or the files in $FILES
  timeout ${TSEL_TIMEOUT:-900} python -m pytest ${FILES:-tests} -m gpu $([ -n "$NOX" ] && echo --maxfail=5 || echo -x) -q --timeout ${TEST_TIMEOUT:-120} --timeout-method=thread ${SEL:+-k "$SEL"} > $OUT/pytest_sel.log 2>&1; tail -15 $OUT/pytest_sel.log | cut -c1-220
}
tests_forced() {   # the Parquet files of the GPU suite with every scan treated as a few-thread task (device-inflated dictionary pages, device-walked run headers, reading task thread)
  COMET_PQ_FEW_THREADS=100000 timeout ${TSEL_TIMEOUT:-900} python -m pytest tests/test_parquet_gpu.py tests/test_parquet_fixtures_gpu.py tests/test_parquet_fuzz_gpu.py tests/test_parquet_page_index_gpu.py tests/test_device_zstd_gpu.py tests/test_device_snappy_gpu.py -m gpu -x -q --timeout ${TEST_TIMEOUT:-120} --timeout-method=thread > $OUT/pytest_forced.log 2>&1; tail -15 $OUT/pytest_forced.log | cut -c1-220
}
smoke() {
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
}
bench() {          # the bench line as the driver runs it
  timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json; echo; tail -3 $OUT/bench.err
}
bench_stats() {    # rocprofv3 kernel statistics of the headline loop only
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $OUT/bench_stats.json 2> /dev/null)
  grep -E '^"k_|utf8_' $OUT/bench_stats/b_kernel_stats.csv | cut -c1-120
}
zstd_variants() {  # the shipped library, then each variant of tools/build_zstd_variants.sh: device parity, the 480-page pipeline, per-kernel averages
  cp datafusion-comet_amd/libcomet.so /tmp/libcomet_shipped.so
  for v in shipped $(ls datafusion-comet_amd/variants 2>/dev/null | sed -n 's/^libcomet_\(.*\)\.so$/\1/p'); do
    if [ $v = shipped ]; then cp /tmp/libcomet_shipped.so datafusion-comet_amd/libcomet.so; else cp datafusion-comet_amd/variants/libcomet_$v.so datafusion-comet_amd/libcomet.so; fi
    case $v in t_*) echo "== $v (timing only: output not checked)";;
      *) timeout 180 python -m pytest tests/test_device_zstd_gpu.py -x -q > $OUT/pytest_zstd_$v.log 2>&1; echo "== $v: $(tail -1 $OUT/pytest_zstd_$v.log | cut -c1-80)"
         timeout 60 python tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --out $OUT/zstd_${v}.json > /dev/null 2> $OUT/zstd_${v}.err; cut -c1-300 $OUT/zstd_${v}.json; echo;;
    esac
    (cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zstd_stats_$v -o z -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --no-check > /dev/null 2>&1)
    grep -E '^"(zs2)' $OUT/zstd_stats_$v/z_kernel_stats.csv | cut -d, -f1,2,4 | tr '\n' ' '; echo
  done
  cp /tmp/libcomet_shipped.so datafusion-comet_amd/libcomet.so
}
zstd_bench() {     # the 480-page zstd pipeline (levels 1 and 3) + per-kernel statistics
  for lvl in 1 3; do
    timeout 90 python tools/snappy_bench.py --codec zstd --level $lvl --pages 480 --kinds decimal_int64,int32_lowcard --out $OUT/zstd_l$lvl.json > /dev/null 2> $OUT/zstd_l$lvl.err; cut -c1-600 $OUT/zstd_l$lvl.json; echo
  done
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zstd_stats -o z -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --codec zstd --level 1 --pages 480 --kinds decimal_int64 --no-check > /dev/null 2>&1)
  grep -E '^"(zs2|sn2|pq_)' $OUT/zstd_stats/z_kernel_stats.csv | cut -c1-120
}
dict_idx_bench() { # the decompression pipelines on the index sections of dictionary-encoded pages (bit-packed random indices: raw or Huffman-only blocks under zstd)
  for C in zstd snappy; do
    timeout 120 python tools/snappy_bench.py --codec $C --level 1 --pages 240 --skip-one-wave --kinds dict_idx_bw4,dict_idx_bw6,dict_idx_bw12 --out $OUT/dict_idx_$C.json > /dev/null 2> $OUT/dict_idx_$C.err; cut -c1-700 $OUT/dict_idx_$C.json; echo
  done
}
snappy_bench() {
  timeout 200 python tools/snappy_bench.py --pages 480 --out $OUT/snappy_bench.json > /dev/null 2> $OUT/snappy_bench.err; cut -c1-900 $OUT/snappy_bench.json; echo
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/snappy_stats -o s -- python $GRAFT_REPO_ROOT/tools/snappy_bench.py --pages 480 --kinds decimal_int64 --no-check --skip-one-wave > /dev/null 2>&1)
  grep -E '^"(sn2|pq_)' $OUT/snappy_stats/s_kernel_stats.csv | cut -c1-120
}
pq_env_sweep() {   # SF10 Q6 from Parquet under environment switches: "$PQ_ENVS" = ;-separated entries "VAR=a,VAR2=b codec:threads:dev"
  IFS=';' read -ra ENTRIES <<< "$PQ_ENVS"
  for E in "${ENTRIES[@]}"; do
    set -- $E
    IFS=: read CODEC THREADS DEV <<< "$2"
    N=sweep_$(echo "$1_$2" | tr -c 'A-Za-z0-9_\n' '_')
    env $(echo $1 | tr ',' ' ') COMET_TRACE_STAGES=1 timeout 240 python tools/parquet_q6.py --codec $CODEC --dir $PQ --scan-threads $THREADS --device-decompress $DEV --steps 6 --out $OUT/$N.json > $OUT/$N.log 2>&1
    echo "== $E: $(python -c "import json;d=json.load(open('$OUT/$N.json'));print('best %.2f ms median %.2f ms on-device pages %s' % (1e3*d['sec_best'],1e3*d['sec_median'],d['pages_decompressed_on_device']))" 2>&1 | tail -1)"
    grep "parquet: \(all launches\|device idle\|scan threads\|column . host\)" $OUT/$N.log | tail -7 | cut -c1-200
  done
}
parquet_q6() {     # SF10 Q6 from Parquet: codec:scan threads (0 = the box's):device decompression (auto / true / false) per entry of $PQ_CFGS; host stage timers in the logs
  for CFG in ${PQ_CFGS:-snappy:0:auto zstd:0:auto zstd:0:false snappy:1:auto zstd:1:auto}; do
    IFS=: read CODEC THREADS DEV <<< "$CFG"
    N=pq6_${CODEC}_t${THREADS}_${DEV}
    COMET_TRACE_STAGES=1 timeout 240 python tools/parquet_q6.py --codec $CODEC --dir $PQ --scan-threads $THREADS --device-decompress $DEV --steps 5 --out $OUT/$N.json > $OUT/$N.log 2>&1
    echo "== $CFG"; cut -c1-420 $OUT/$N.json; echo; grep "parquet:" $OUT/$N.log | tail -${PQ_TRACE_LINES:-14} | cut -c1-230
  done
}
pq_timeline() {    # device timeline (kernels + copies) of one SF10 Q6 scan per configuration in $PQ_CFGS
  for CFG in ${PQ_CFGS:-snappy:0:auto zstd:0:auto}; do
    IFS=: read CODEC THREADS DEV <<< "$CFG"
    N=tl_${CODEC}_t${THREADS}_${DEV}
    (cd /tmp && timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$N -o t -- python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec $CODEC --dir $PQ --scan-threads $THREADS --device-decompress $DEV --steps 2 > $OUT/$N.log 2>&1)
    python tools/timeline.py $(find $OUT/$N -name "*kernel_trace.csv") $(find $OUT/$N -name "*memory_copy_trace.csv") > $OUT/$N.txt 2>&1
    echo "== $CFG"; head -${TL_LINES:-70} $OUT/$N.txt | cut -c1-200
  done
}
q3_stats() {       # SF100 Q3 on one GPU: kernel statistics + PMC of the probes and the filter (separate passes)
  Q3="python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 3 --warmup 1 --no-verify"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q3_stats -o q3 -- $Q3 > $OUT/q3_stats.log 2>&1
  grep -E '^"k_|^"comet|^"exch' $OUT/q3_stats/q3_kernel_stats.csv | cut -c1-120
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q3_fetch -o q3 -- $Q3 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/q3_tcc -o q3 -- $Q3 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/q3_sq -o q3 -- $Q3 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_join_summary.py $OUT q3 > $OUT/q3_probe_pmc.txt 2>&1; head -60 $OUT/q3_probe_pmc.txt | cut -c1-260
}
q3_timeline() {    # device timeline of one SF100 Q3 run (kernels + copies in start order): where the Final aggregate and the top-10 spend their time
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/q3_tl -o t -- python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 2 --warmup 1 --no-verify > $OUT/q3_tl.log 2>&1)
  python tools/timeline.py $(find $OUT/q3_tl -name "*kernel_trace.csv") $(find $OUT/q3_tl -name "*memory_copy_trace.csv") > $OUT/q3_tl.txt 2>&1
  head -${TL_LINES:-200} $OUT/q3_tl.txt | cut -c1-160
}
q3_host() {        # host calls and device work of the last SF100 Q3 run side by side: what fills the gaps between the kernels
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/q3_host -o h -- python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 2 --warmup 1 --no-verify > $OUT/q3_host.log 2>&1)
  python tools/hip_timeline.py $OUT/q3_host k_filter 4 > $OUT/q3_host.txt 2>&1; rm -rf $OUT/q3_host; head -${TL_LINES:-400} $OUT/q3_host.txt | cut -c1-140
}
q3() {
  timeout 300 python tools/q3_dist.py --orders 150000000 --steps 5 --warmup 2 --out $OUT/q3.json > $OUT/q3.log 2>&1; cut -c1-700 $OUT/q3.json; echo
}
q95() {
  timeout 300 python tools/q95_bench.py --orders 16000000 --reps 3 > $OUT/q95.json 2> $OUT/q95.err; cut -c1-600 $OUT/q95.json; echo
}
executor() {       # 8 / 16 concurrent plans, one host thread and one scan thread each
  timeout 600 python tools/executor_bench.py --dir $PQ --steps 2 ${EXEC_ARGS:---busy} --out $OUT/executor.json > $OUT/executor.log 2>&1; python -c "
import json;d=json.load(open('$OUT/executor.json'));print('link', d.get('pcie_link_GBps_measured'))
for k,v in d['legs'].items():
  for t,e in v.items():
    if t.startswith('tasks_'): print(k,t,{a:(round(b,3) if isinstance(b,float) else b) for a,b in e.items()})" 2>&1 | tail -12; tail -3 $OUT/executor.log | cut -c1-300
}
executor_envs() {  # the executor leg under environment switches: "$EXEC_ENVS" = ;-separated "VAR=a,VAR2=b" entries ("-" = none); legs in $EXEC_LEGS
  IFS=';' read -ra ENTRIES <<< "${EXEC_ENVS:--}"
  for E in "${ENTRIES[@]}"; do
    N=exec_$(echo "$E" | tr -c 'A-Za-z0-9_\n' '_')
    env $([ "$E" = "-" ] || echo $E | tr ',' ' ') ${EXEC_TRACE:+COMET_TRACE_STAGES=1} timeout 300 python tools/executor_bench.py --dir $PQ --steps ${EXEC_STEPS:-3} --no-link --legs $(echo ${EXEC_LEGS:-parquet_snappy,parquet_zstd} | tr ' ' ',') --tasks ${EXEC_TASKS:-8,16} --out $OUT/$N.json > $OUT/$N.log 2>&1
    echo "== $E: $(python -c "
import json;d=json.load(open('$OUT/$N.json'))
print('  '.join(f'{k}/{t}: {e[\"wall_ms\"]:.1f} ms (median {e[\"wall_ms_median\"]:.1f})' for k,v in d['legs'].items() for t,e in v.items() if t.startswith('tasks_')))" 2>&1 | tail -1)  slow copies: $(grep -c 'held this thread' $OUT/$N.log), longest $(grep 'held this thread' $OUT/$N.log | sed 's/.*held this thread \([0-9.]*\) ms.*/\1/' | sort -n | tail -1) ms"
  done
}
executor_trace() { # one traced wave of 8 one-core tasks per codec: every task's stage lines (times relative to its own scan's start)
  for L in ${EXEC_LEGS:-parquet_snappy parquet_zstd}; do
    COMET_TRACE_STAGES=1 timeout 300 python tools/executor_bench.py --dir $PQ --steps 2 --no-link --tasks 8 --legs $L > /dev/null 2> $OUT/exec_trace_$L.log
    echo "== $L: $(grep -c 'device idle' $OUT/exec_trace_$L.log) scans traced, $(grep -c 'held this thread' $OUT/exec_trace_$L.log) slow copies"; grep "device idle\|held this thread\|launch took" $OUT/exec_trace_$L.log | tail -${TRACE_LINES:-30} | cut -c1-200
  done
}
executor_wave() {  # one wave of $WAVE_TASKS one-core tasks per leg under rocprofv3 (kernels, copies, HIP API): tools/wave_timeline.py says who waits for whom
  for L in ${EXEC_LEGS:-parquet_snappy parquet_zstd}; do
    (cd /tmp && COMET_TRACE_STAGES=1 timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT/wave_$L -o w -- python $GRAFT_REPO_ROOT/tools/executor_bench.py --dir $PQ --steps 1 --no-link --tasks ${WAVE_TASKS:-8} --legs $L > /dev/null 2> $OUT/wave_$L.log)
    python tools/wave_timeline.py $OUT/wave_$L ${WAVE_MIN_US:-300} $OUT/wave_${L}_device.txt > $OUT/wave_$L.txt 2>&1; rm -rf $OUT/wave_$L
    echo "== $L"; head -${WAVE_LINES:-60} $OUT/wave_$L.txt | cut -c1-160
  done
}
q95_stats() {      # kernel statistics of TPC-DS Q95 stage A on one GPU
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats -o q95 -- python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 2 --verify none > $OUT/q95_stats.log 2>&1)
  python - <<PYEOF
import csv,re
rows=list(csv.DictReader(open("$OUT/q95_stats/q95_kernel_stats.csv")))
for r in rows[:14]:
    m=re.search(r"(k_\w+|[a-z_0-9]+_kernel|__amd\w+)", r["Name"]); print(f'{(m.group(1) if m else r["Name"][:30]):28s} calls {r["Calls"]:>4s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_ms {float(r["AverageNs"])/1e6:8.3f}')
PYEOF
}
q6_host() {        # host calls and device work of the last SF10 Q6 task (HBM-resident): what the task costs besides its kernel
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT/q6_host -o h -- python $GRAFT_REPO_ROOT/tools/resident.py --query q6 --rows 59986052 --steps 6 --no-check > $OUT/q6_host.log 2>&1)
  python tools/hip_timeline.py $OUT/q6_host k_agg ${TL_MIN_US:-2} 1 > $OUT/q6_host.txt 2>&1; rm -rf $OUT/q6_host; tail -${TL_LINES:-90} $OUT/q6_host.txt | cut -c1-140
}
q95_host() {       # host calls and device work of the last Q95 run side by side: what fills the gaps between the kernels
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT/q95_host -o h -- python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 2 --verify none > $OUT/q95_host.log 2>&1)
  python tools/hip_timeline.py $OUT/q95_host k_filter ${TL_MIN_US:-4} ${Q95_TL_NTH:-3} > $OUT/q95_host.txt 2>&1; rm -rf $OUT/q95_host; tail -${TL_LINES:-60} $OUT/q95_host.txt | cut -c1-140
}
q95_pmc() {        # per-kernel HBM traffic and wait cycles of Q95 stage A (separate PMC passes, as the guide prescribes)
  Q95="python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 1 --verify none"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats -o q95 -- $Q95 > $OUT/q95_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q95_fetch -o q95 -- $Q95 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/q95_tcc -o q95 -- $Q95 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/q95_sq -o q95 -- $Q95 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_join_summary.py $OUT q95 > $OUT/q95_join_pmc.txt 2>&1; head -60 $OUT/q95_join_pmc.txt | cut -c1-260
}
pmc_sizes() {      # HBM bytes from the size-classed TCC->EA request counters (tools/pmc_sizes.py): calibration on a known 4 GiB read, then Q95, Q3 and the headline's kernels
  timeout 400 python tools/pmc_sizes.py $OUT/pmc_sizes_calib.txt --calib -- python $GRAFT_REPO_ROOT/tools/pmc_calibrate.py --run | cut -c1-200
  timeout 500 python tools/pmc_sizes.py $OUT/pmc_sizes_q95.txt -- python $GRAFT_REPO_ROOT/tools/q95_bench.py --orders 16000000 --reps 1 --verify none | cut -c1-200
  timeout 500 python tools/pmc_sizes.py $OUT/pmc_sizes_q3.txt -- python $GRAFT_REPO_ROOT/tools/q3_dist.py --orders 150000000 --steps 1 --warmup 1 --no-verify | head -24 | cut -c1-200
  timeout 500 python tools/pmc_sizes.py $OUT/pmc_sizes_q1.txt -- python $GRAFT_REPO_ROOT/tools/resident.py --rows 600037902 --query q1,q6 --steps 3 --no-check | head -12 | cut -c1-200
}
q95_metrics() {    # Q95 stage A run by run: which table each join took, per-kernel event times
  timeout 300 python tools/q95_metrics.py 2> $OUT/q95_metrics.err | tee -a $OUT/q95_metrics.jsonl | cut -c1-700 || tail -5 $OUT/q95_metrics.err
}
q95_jit_ab() {     # does a code object compiled under rocprofv3 differ from one compiled without it?  (a20: k_jprobe_b 2.50 ms under the profiler on a fresh box, 2.94 elsewhere)
  mkdir -p /tmp/cacheA /tmp/cacheB
  (cd /tmp && COMET_JIT_CACHE_DIR=/tmp/cacheA timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jitab -o a -- python $GRAFT_REPO_ROOT/tools/q95_metrics.py --runs 3 2> /dev/null | cut -c1-330)
  echo "-- plain, own cache"; COMET_JIT_CACHE_DIR=/tmp/cacheB timeout 300 python tools/q95_metrics.py --runs 3 2> /dev/null | cut -c1-330
  echo "-- plain, the profiler-compiled cache"; COMET_JIT_CACHE_DIR=/tmp/cacheA timeout 300 python tools/q95_metrics.py --runs 3 2> /dev/null | cut -c1-330
  echo "-- under the profiler, the plain-compiled cache"; (cd /tmp && COMET_JIT_CACHE_DIR=/tmp/cacheB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jitab2 -o a -- python $GRAFT_REPO_ROOT/tools/q95_metrics.py --runs 3 2> /dev/null | cut -c1-330)
  (cd /tmp/cacheA && md5sum *.hsaco | sort -k2) > $OUT/cacheA.md5; (cd /tmp/cacheB && md5sum *.hsaco | sort -k2) > $OUT/cacheB.md5
  diff $OUT/cacheA.md5 $OUT/cacheB.md5 | head -20
  mkdir -p $OUT/hsaco; for f in $(diff $OUT/cacheA.md5 $OUT/cacheB.md5 | grep '^<' | awk '{print $3}' | head -4); do cp /tmp/cacheA/$f $OUT/hsaco/A_$f; cp /tmp/cacheB/$f $OUT/hsaco/B_$f; done; ls -la $OUT/hsaco | head
}
headline_ab() {    # the headline loop (SF100 Q1, HBM-resident) under environment configurations ("$HL_CFGS", as q95_cfgs; default: the torch wheel's compiler against the installed one), on ONE box
  IFS='|' read -ra ENTRIES <<< "${HL_CFGS:-COMET_SYSTEM_COMGR=0|-|COMET_SYSTEM_COMGR=0|-}"
  for E in "${ENTRIES[@]}"; do
    E=$(echo $E)
    [ "$E" = "-" ] && E=""
    env $E timeout 600 python bench.py --steps ${HL_STEPS:-20} --warmup 3 --no-extra-legs --no-cpu-baseline 2> /dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f'{d[\"value\"]/1e9:7.2f} G rows/s {d[\"ms_per_step\"]:7.3f} ms  frac {d[\"roofline\"][\"frac\"]}  kernel_ms {d[\"roofline\"].get(\"kernel_ms\")}  $E  ', d['config'].get('jit_toolchain'))"
  done
}
fused_check() {    # tools/fused_build_check.py under environment configurations ("$FC_CFGS", as q95_cfgs), arguments "$FC_ARGS" (fuseBuild modes)
  IFS='|' read -ra ENTRIES <<< "${FC_CFGS:--}"
  for E in "${ENTRIES[@]}"; do
    E=$(echo $E)
    [ "$E" = "-" ] && E=""
    echo "== $E"; env $E timeout 300 python tools/fused_build_check.py ${FC_ARGS:-always false} 2>&1 | grep -v "amdgpu.ids" | tail -14 | cut -c1-260
  done
}
strtod_check() {   # tools/strtod_check.py under environment configurations ("$SD_CFGS", as q95_cfgs)
  IFS='|' read -ra ENTRIES <<< "${SD_CFGS:--|COMET_SYSTEM_COMGR=0}"
  for E in "${ENTRIES[@]}"; do
    E=$(echo $E)
    [ "$E" = "-" ] && E=""
    echo "== $E"; env $E timeout 300 python tools/strtod_check.py 2>&1 | grep -v amdgpu.ids | sed -n ${SD_LINES:-1,2p} | cut -c1-300
  done
}
q95_cfgs() {       # Q95 stage A under environment configurations: "$Q95_CFGS" = |-separated entries, each a space-separated list of VAR=value ("-" = none)
  IFS='|' read -ra ENTRIES <<< "${Q95_CFGS:--}"
  for E in "${ENTRIES[@]}"; do
    E=$(echo $E)
    [ "$E" = "-" ] && E=""
    env $E timeout 300 python tools/q95_variant.py --root . --reps ${Q95_REPS:-3} --tag "$E" 2> $OUT/q95_cfg.err | tee -a $OUT/q95_cfgs.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f'{min(d[\"stage_a_ms\"]):7.2f} ms  {d[\"tag\"]}  {d[\"result\"]}')" || tail -3 $OUT/q95_cfg.err
  done
}
q3_cfgs() {        # SF100 Q3 on one GPU under environment configurations ("$Q3_CFGS", as q95_cfgs)
  IFS='|' read -ra ENTRIES <<< "${Q3_CFGS:--}"
  for E in "${ENTRIES[@]}"; do
    E=$(echo $E)
    [ "$E" = "-" ] && E=""
    env $E timeout 300 python tools/q3_dist.py --orders 150000000 --steps 4 --warmup 2 --kernel-times > $OUT/q3_cfg.json 2> $OUT/q3_cfg.err; python -c "
import json; d=json.load(open('$OUT/q3_cfg.json')); k=(d.get('roofline') or {}).get('kernels') or []
print(f'{d[\"sec_per_run\"]*1e3:7.3f} ms  $E  stages {d[\"stage_ms_rank0\"]}  ' + ' '.join(f'{x[\"name\"]}={x[\"ms\"]:.2f}' for x in k[:6]), 'verified', d.get('verified_vs_torch'))" 2>&1 | tail -1
    echo "{\"cfg\": \"$E\", \"out\": $(cat $OUT/q3_cfg.json)}" >> $OUT/q3_cfgs.jsonl
  done
}
q95_bisect() {     # TPC-DS Q95 stage A with the libraries of earlier commits (bisect/<tag>/, built from git worktrees) and with HEAD's join switches, on ONE box
  for v in $(ls bisect 2>/dev/null); do timeout 200 python tools/q95_variant.py --root bisect/$v --tag $v 2> $OUT/q95_$v.err | tee -a $OUT/q95_bisect.jsonl | cut -c1-300; done
  timeout 200 python tools/q95_variant.py --root . --tag head 2> $OUT/q95_head.err | tee -a $OUT/q95_bisect.jsonl | cut -c1-300
  for E in ${Q95_ENVS:-COMET_JOIN_KEYMAP=0 COMET_JOIN_DIRECT=0 COMET_JOIN_COUNT_RUNS=0}; do
    env $E timeout 200 python tools/q95_variant.py --root . --tag "head $E" 2> $OUT/q95_env.err | tee -a $OUT/q95_bisect.jsonl | cut -c1-300
  done
  for v in ${Q95_STATS:-r3 head}; do
    R=bisect/$v; [ $v = head ] && R=.
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q95_stats_$v -o q -- python $GRAFT_REPO_ROOT/tools/q95_variant.py --root $GRAFT_REPO_ROOT/$R --reps 2 > /dev/null 2>&1)
    echo "== kernel stats $v (3 runs)"; python tools/kernel_stats_csv.py $OUT/q95_stats_$v/q_kernel_stats.csv 14 2>&1 | cut -c1-150 | tee $OUT/q95_kernel_stats_$v.txt
  done
}
small_legs() {     # the bench legs' tools at toy sizes: a typo must not cost a full-size run
  timeout 200 python tools/q3_dist.py --orders 1500000 --steps 1 --warmup 1 --kernel-times 2>&1 | tail -1 | cut -c1-1500
  timeout 200 python tools/q95_dist.py --orders 200000 --steps 1 --warmup 1 --kernel-times 2>&1 | tail -1 | cut -c1-1500
}
read_probe() {     # page cache -> pinned memory -> device, nothing else: what bounds a scan before the GPU sees a byte
  ls $PQ/*.parquet > /dev/null 2>&1 || timeout 200 python tools/parquet_q6.py --codec snappy --dir $PQ --steps 1 > /dev/null 2>&1
  timeout 200 python tools/read_probe.py --file $(ls -S $PQ/*.parquet | head -1) > $OUT/read_probe.json 2> $OUT/read_probe.err; cat $OUT/read_probe.json; tail -2 $OUT/read_probe.err
}
probes() {         # the box itself: HBM streaming rates, PCIe
  timeout 200 python tools/hbm_probe.py > $OUT/hbm_probe.json 2>/dev/null; cut -c1-400 $OUT/hbm_probe.json; echo
  timeout 200 python tools/h2d_probe.py > $OUT/h2d.json 2>/dev/null; cut -c1-400 $OUT/h2d.json; echo
}

for s in "$@"; do
  echo "#### $s"; t0=$(date +%s); $s; echo "#### $s done in $(( $(date +%s) - t0 )) s"
done
find $OUT -name "*.csv" -size +2M -delete
