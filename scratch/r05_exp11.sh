#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp11; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -n 4 > $O/gpu_tests.txt 2>&1
tail -n 6 $O/gpu_tests.txt
for i in 1 2 3 4 5 6; do ROWS=b1,b4,104x152,256x256 python scratch/stall_fresh.py; ROWS=104x152,112x144,96x168,152x104 python scratch/stall_fresh.py; done > $O/fresh.txt 2>&1
grep -v amdgpu $O/fresh.txt
python scratch/size_sweep.py 128x128 64x64 104x152 152x104 112x144 144x112 96x168 168x96 80x192 192x80 96x96 160x160 120x120 136x104 112x112 80x80 256x256 > $O/sizes.txt 2>&1
tail -n 30 $O/sizes.txt
