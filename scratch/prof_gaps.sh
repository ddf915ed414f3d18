#!/bin/bash
# kernel trace with durations and the gaps between dispatches: scratch/prof_gaps.sh <tag> <script.py>  ->  gpurun_out/<tag>/gaps.md
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=$1; shift
O=gpurun_out/$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python "$@" > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_trace.md 2>> $O/err.txt
python tools/rocpd_gaps.py $DB > $O/gaps.md 2>> $O/err.txt
rm -rf $O/trace
cut -c1-200 $O/gaps.md
