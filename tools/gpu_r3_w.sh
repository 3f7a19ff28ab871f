#!/bin/bash
# PMC view of the snappy pipeline's kernels (480 pages of 1 MiB, decimal-as-INT64)
cd /tmp
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/snappy_bench.py --pages 480 --kinds decimal_int64 --skip-one-wave --no-check"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq -o p -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $CMD > /dev/null 2>&1
for d in sq sq2; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$d "(anonymous namespace)::sn2"; done
find $OUT -name "*.csv" -size +3M -delete
