#!/bin/bash
# issue / LDS counters of the generate pass (scratch/pipe_time.py: 2048 planes of 128 x 128 through sonar_power_irfft2_f32), two passes
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/r03_pipe
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_a -o p -- python scratch/pipe_time.py > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAIT_ANY -d $O/pmc_b -o p -- python scratch/pipe_time.py > $O/pmc_b.log 2>&1
for p in a b; do python tools/rocpd_pmc.py $(find $O/pmc_$p -name "*.db" | head -1) power > $O/pmc_$p.txt 2>> $O/err.txt; done
rm -rf $O/pmc_a $O/pmc_b
cat $O/pmc_a.txt $O/pmc_b.txt
