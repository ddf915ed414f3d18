#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp7; mkdir -p $O
V=""
for v in v3 v4 v5 v6 v6_44 v6_43 v6_34 v6_p11 v6_25 v6_24 v6_33; do V="$V scratch/bin/pwvar/lib_$v.so"; done
MODE=ahead python scratch/pipe_ab.py $V > $O/ab_ahead.txt 2>&1
MODE=final python scratch/pipe_ab.py $V > $O/ab_final.txt 2>&1
tail -n 40 $O/ab_ahead.txt $O/ab_final.txt
