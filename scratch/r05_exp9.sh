#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp9; mkdir -p $O
V=""
for v in v3 v6 v7 v7_33 v7_24 v7_34 v7_44; do V="$V scratch/bin/pwvar/lib_$v.so"; done
python scratch/pw_verify.py scratch/bin/pwvar/lib_v7.so > $O/verify.txt 2>&1
MODE=ahead python scratch/pipe_ab.py $V > $O/ab_ahead.txt 2>&1
MODE=final python scratch/pipe_ab.py $V > $O/ab_final.txt 2>&1
SONAR_TRACE_AHEAD=1 python scratch/pipe_trace.py scratch/bin/pwvar/lib_v7t.so > $O/trace_ahead.txt 2>&1
tail -n 40 $O/verify.txt $O/ab_ahead.txt $O/ab_final.txt $O/trace_ahead.txt
