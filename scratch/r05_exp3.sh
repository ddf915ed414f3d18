#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_exp3; mkdir -p $O
for v in trace35 trace44; do
  python scratch/pipe_trace.py scratch/bin/pwvar/lib_$v.so > $O/${v}_final.txt 2>&1
  SONAR_TRACE_AHEAD=1 python scratch/pipe_trace.py scratch/bin/pwvar/lib_$v.so > $O/${v}_ahead.txt 2>&1
done
tail -n 15 $O/*.txt
