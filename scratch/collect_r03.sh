#!/bin/bash
# gpurun_out/r03 (scratch/prof_r03.sh) + gpurun_out/r03_sizes.txt (scratch/size_sweep.py) -> profiles/r03_*
cd "$(dirname "$0")/.."
O=gpurun_out/r03
python tools/make_traffic_json.py $O/traffic_raw.json > profiles/r03_traffic.json
cp $O/bench.json profiles/r03_bench_line.json
cp $O/kernel_trace.md profiles/r03_bench_kernel_trace.md
cp $O/pmc_sq_a.txt profiles/r03_pmc_issue_a.txt
cp $O/pmc_sq_b.txt profiles/r03_pmc_issue_b.txt
cp gpurun_out/r03_sizes.txt profiles/r03_sizes.txt
{ grep -v amdgpu $O/pair_time.txt; echo "final pass alone, then a sampler's steady state (MODE=ahead):"; grep -v amdgpu $O/pipe_time.txt; } > profiles/r03_pair_time.txt
{ echo "== final pass alone (sonar_power_irfft2_f32, z = NULL) =="; grep -v amdgpu $O/pipe_phases_plain.txt
  echo "== with the next call's statistics (sonar_power_noise_ahead_f32; the trace build's stamps cost ~10 % of the kernel) =="; grep -v amdgpu $O/pipe_phases_ahead.txt; } > profiles/r03_power_phases.txt
