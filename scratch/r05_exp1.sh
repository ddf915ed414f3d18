#!/bin/bash
# round 5, first GPU pass: instruction costs, correctness and launch time of the draw / row-pass variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_exp1
O=gpurun_out/r05_exp1
./scratch/bin/valu_rate > $O/valu_rate.txt 2>&1
V="scratch/bin/pwvar/lib_r04.so scratch/bin/pwvar/lib_v1full.so scratch/bin/pwvar/lib_v2.so scratch/bin/pwvar/lib_v2_24.so scratch/bin/pwvar/lib_v2_34.so scratch/bin/pwvar/lib_v2_25.so scratch/bin/pwvar/lib_v2_44.so"
python scratch/pw_verify.py scratch/bin/pwvar/lib_v1full.so scratch/bin/pwvar/lib_v2.so scratch/bin/pwvar/lib_v2_24.so > $O/verify.txt 2>&1
python scratch/pipe_time.py $V > $O/time_final.txt 2>&1
MODE=ahead python scratch/pipe_time.py $V > $O/time_ahead.txt 2>&1
tail -n 40 $O/verify.txt $O/time_final.txt $O/time_ahead.txt
