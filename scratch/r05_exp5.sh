#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp5; mkdir -p $O
V=""
for v in v3 v4 v4_44 v5 v5_44 v5_p11 v5_43 v5_34; do V="$V scratch/bin/pwvar/lib_$v.so"; done
python scratch/pw_verify.py scratch/bin/pwvar/lib_v5.so > $O/verify.txt 2>&1
python scratch/pipe_time.py $V > $O/time_final.txt 2>&1
MODE=ahead python scratch/pipe_time.py $V > $O/time_ahead.txt 2>&1
python scratch/pipe_trace.py scratch/bin/pwvar/lib_v5t.so > $O/trace_final.txt 2>&1
SONAR_TRACE_AHEAD=1 python scratch/pipe_trace.py scratch/bin/pwvar/lib_v5t.so > $O/trace_ahead.txt 2>&1
tail -n 40 $O/verify.txt $O/time_final.txt $O/time_ahead.txt $O/trace_final.txt $O/trace_ahead.txt
