#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp4; mkdir -p $O
V=""
for v in v3 v3_44 p00 p11 p10 p01 p30 c43 c35 c53; do V="$V scratch/bin/pwvar/lib_$v.so"; done
python scratch/pipe_time.py $V > $O/time_final.txt 2>&1
MODE=ahead python scratch/pipe_time.py $V > $O/time_ahead.txt 2>&1
tail -n 40 $O/time_final.txt $O/time_ahead.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
tail -n 15 $O/gpu_tests.txt
