#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp2; mkdir -p $O
V="scratch/bin/pwvar/lib_r04.so scratch/bin/pwvar/lib_v2.so scratch/bin/pwvar/lib_v3.so scratch/bin/pwvar/lib_v3_44.so scratch/bin/pwvar/lib_v3_26.so scratch/bin/pwvar/lib_v3_34.so scratch/bin/pwvar/lib_v3_24.so"
python scratch/pw_verify.py scratch/bin/pwvar/lib_v3.so scratch/bin/pwvar/lib_v3_24.so > $O/verify.txt 2>&1
python scratch/pipe_time.py $V > $O/time_final.txt 2>&1
MODE=ahead python scratch/pipe_time.py $V > $O/time_ahead.txt 2>&1
tail -n 40 $O/verify.txt $O/time_final.txt $O/time_ahead.txt
