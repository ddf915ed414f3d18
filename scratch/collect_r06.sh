#!/bin/bash
# gpurun_out/r06 (scratch/prof_r06.sh) + gpurun_out/r06_sizes.txt (scratch/size_sweep.py) -> profiles/r06_*
cd "$(dirname "$0")/.."
O=gpurun_out/r06
python tools/make_traffic_json.py $O/traffic_raw.json > $O/traffic.json && python -c "import json,sys; json.load(open(sys.argv[1]))" $O/traffic.json \
  && cp $O/traffic.json profiles/r06_traffic.json || echo "traffic table NOT updated (tools/make_traffic_json.py failed)"
cp $O/bench.json profiles/r06_bench_line.json
cp $O/kernel_trace.md profiles/r06_bench_kernel_trace.md
cp $O/pmc_sq_a.txt profiles/r06_pmc_issue_a.txt
cp $O/pmc_sq_b.txt profiles/r06_pmc_issue_b.txt
[ -f gpurun_out/r06_sizes.txt ] && cp gpurun_out/r06_sizes.txt profiles/r06_sizes.txt
{ grep -v amdgpu $O/pair_time.txt; echo "final pass alone, then a sampler's steady state (MODE=ahead):"; grep -v amdgpu $O/pipe_time.txt; } > profiles/r06_pair_time.txt
{ echo "== final pass alone (sonar_power_irfft2_f32, z = NULL) =="; grep -v amdgpu $O/pipe_phases_plain.txt
  echo "== with the next call's statistics (sonar_power_noise_ahead_f32; the trace build's stamps cost ~10 % of the kernel) =="; grep -v amdgpu $O/pipe_phases_ahead.txt; } > profiles/r06_power_phases.txt
for f in fill_rates fill_ahead brownian_tree pyramid_ahead sizes_sampler lowpass; do
  [ -f $O/$f.txt ] && grep -v amdgpu $O/$f.txt > profiles/r06_$f.txt
done
for f in sizes_split perlin_parts pipe_scaling; do
  [ -f $O/$f.txt ] && grep -v amdgpu $O/$f.txt > profiles/r06_$f.txt
done
