#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp12; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -n 4 -k "general or size or bucket or sdxl or spectral or block or plan" > $O/gpu_tests.txt 2>&1
tail -n 4 $O/gpu_tests.txt
python scratch/size_sweep.py 128x128 104x152 152x104 112x144 144x112 96x168 168x96 80x192 192x80 96x96 > $O/sizes.txt 2>&1
cat $O/sizes.txt | grep -v amdgpu
python scratch/fuzz_spectral.py 120 5 > $O/fuzz.txt 2>&1; tail -n 3 $O/fuzz.txt
