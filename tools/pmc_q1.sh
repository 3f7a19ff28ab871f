#!/bin/bash
# PMC passes for the grouped-aggregate kernel (run on the GPU box via gpurun). Usage: tools/pmc_q1.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
P="python $GRAFT_REPO_ROOT/tools/resident.py --query q1 --steps 2 --no-check"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/p1 -o q1 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p2 -o q1 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/p3 -o q1 -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p4 -o q1 -- $P > /dev/null 2>&1
ls $OUT/*
