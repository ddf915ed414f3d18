#!/bin/bash
# round-1 profile set: bench line, kernel trace + stats, PMC (FETCH_SIZE and WRITE_SIZE in separate passes) for bench and calibration
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
python bench.py > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r01/trace -o t -- python bench.py --no-cpu-baseline > gpurun_out/r01/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/r01/pmc_bench_$c -o p -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/r01/pmc_calib_$c -o p -- python scratch/calib.py > /dev/null 2>&1
done
find gpurun_out/r01 -name "*.db" | head -20
cat gpurun_out/r01/bench.json
