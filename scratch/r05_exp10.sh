#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp10; mkdir -p $O
V=""
for v in v3 v7 v8 v8_33 v8_24; do V="$V scratch/bin/pwvar/lib_$v.so"; done
python scratch/pw_verify.py scratch/bin/pwvar/lib_v8.so > $O/verify.txt 2>&1
MODE=ahead python scratch/pipe_ab.py $V > $O/ab_ahead.txt 2>&1
MODE=final python scratch/pipe_ab.py $V > $O/ab_final.txt 2>&1
tail -n 40 $O/verify.txt $O/ab_ahead.txt $O/ab_final.txt
