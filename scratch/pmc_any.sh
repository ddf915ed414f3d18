#!/bin/bash
# SQ issue / LDS counters of the power kernels of any script: scratch/pmc_any.sh <tag> <script.py> [args]  ->  gpurun_out/<tag>/pmc_{a,b}.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=$1; shift
O=gpurun_out/$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/a -o p -- python "$@" > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAIT_ANY -d $O/b -o p -- python "$@" > $O/b.log 2>&1
for p in a b; do python tools/rocpd_pmc.py $(find $O/$p -name "*.db" | head -1) power > $O/pmc_$p.txt 2>> $O/err.txt; done
rm -rf $O/a $O/b
cat $O/pmc_a.txt $O/pmc_b.txt
