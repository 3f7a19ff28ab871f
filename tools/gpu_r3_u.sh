#!/bin/bash
# PMC view of the decode kernel (SF10 Q6 file without compression: the decode kernels alone matter)
cd /tmp
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3u
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/parquet_q6.py --codec none --steps 3"
timeout 300 $CMD > /dev/null 2>&1    # writes the file
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq -o p -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/wr -o p -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $CMD > /dev/null 2>&1
for d in sq fetch wr sq2; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$d pq_decode_runs; done
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3u/sq/**/*kernel_trace.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"].startswith("pq_decode_runs"):
        print(r["Kernel_Name"][:30], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"], r["Scratch_Size"])
PY
find $OUT -name "*.csv" -size +3M -delete
