#!/bin/bash
# kernel trace of any script: scratch/prof_any.sh <tag> <script.py> [args]  ->  gpurun_out/<tag>/kernel_trace.md
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=$1; shift
O=gpurun_out/$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python "$@" > $O/trace.log 2>&1
python tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_trace.md 2>> $O/err.txt
rm -rf $O/trace
cut -c1-180 $O/kernel_trace.md
