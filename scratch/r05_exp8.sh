#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp8; mkdir -p $O
V=""
for v in v6 s1 s2 s3 s4 s8 s16 s28 s30 s31; do V="$V scratch/bin/pwvar/lib_$v.so"; done
ROUNDS=5 MODE=final python scratch/pipe_ab.py $V > $O/ab_final.txt 2>&1
tail -n 40 $O/ab_final.txt
