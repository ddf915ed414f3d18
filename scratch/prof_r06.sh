#!/bin/bash
# round-6 profile set: bench line, kernel trace + stats of the same command, PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
# over scratch/prof_workload.py, issue / LDS counters of the pipelined power-noise kernel in a sampler's steady state (two more passes),
# and its phase timeline (trace build: scratch/pwv.sh trace "-DSONAR_PW_TRACE" first)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/r06
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --no-cpu-baseline > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python scratch/prof_workload.py > $O/pmc_$c.log 2>&1
done
export MODE=ahead
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_sq_a -o p -- python scratch/pipe_time.py > $O/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAIT_ANY -d $O/pmc_sq_b -o p -- python scratch/pipe_time.py > $O/pmc_sq_b.log 2>&1
unset MODE
python tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_trace.md 2>> $O/bench.err
python tools/rocpd_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) > $O/traffic_raw.json 2>> $O/bench.err
for p in a b; do python tools/rocpd_pmc.py $(find $O/pmc_sq_$p -name "*.db" | head -1) power > $O/pmc_sq_$p.txt 2>> $O/bench.err; done
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq_a $O/pmc_sq_b
if [ -f scratch/bin/pwvar/lib_trace.so ]; then
  python scratch/pipe_trace.py > $O/pipe_phases_plain.txt 2>&1
  SONAR_TRACE_AHEAD=1 python scratch/pipe_trace.py > $O/pipe_phases_ahead.txt 2>&1
fi
python scratch/pipe_time.py > $O/pipe_time.txt 2>&1
MODE=ahead python scratch/pipe_time.py >> $O/pipe_time.txt 2>&1
python scratch/pair_time.py > $O/pair_time.txt 2>&1
python scratch/fill_rates.py > $O/fill_rates.txt 2>&1
python scratch/fill_ahead_dbg.py > $O/fill_ahead.txt 2>&1
python scratch/brownian_tree_time.py > $O/brownian_tree.txt 2>&1
python scratch/pyramid_ahead_time.py > $O/pyramid_ahead.txt 2>&1
python scratch/sizes_sampler.py > $O/sizes_sampler.txt 2>&1
python scratch/lowpass_time.py > $O/lowpass.txt 2>&1
ls -la $O; tail -c 400 $O/bench.json
