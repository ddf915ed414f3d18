#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp6; mkdir -p $O
V=""
for v in v3 v4 v5 v6 v6_44 v6_43 v6_34 v6_p11; do V="$V scratch/bin/pwvar/lib_$v.so"; done
python scratch/pw_verify.py scratch/bin/pwvar/lib_v6.so > $O/verify.txt 2>&1
python scratch/pipe_time.py $V > $O/time_final.txt 2>&1
MODE=ahead python scratch/pipe_time.py $V > $O/time_ahead.txt 2>&1
python scratch/pipe_trace.py scratch/bin/pwvar/lib_v6t.so > $O/trace_final.txt 2>&1
SONAR_TRACE_AHEAD=1 python scratch/pipe_trace.py scratch/bin/pwvar/lib_v6t.so > $O/trace_ahead.txt 2>&1
tail -n 40 $O/verify.txt $O/time_final.txt $O/time_ahead.txt $O/trace_final.txt $O/trace_ahead.txt
