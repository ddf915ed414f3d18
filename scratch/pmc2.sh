#!/bin/bash
# usage: scratch/pmc2.sh <tag> <binary>
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/pmc_${tag}_a -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_${tag}_b -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT -d gpurun_out/pmc_${tag}_c -o p -- "$@" > /dev/null 2>&1
