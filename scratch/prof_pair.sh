#!/bin/bash
# kernel trace of the normalised power-noise launch pair (scratch/pw_kern.py: 512 SDXL latents, generate mode)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/r03_pair
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python scratch/pw_kern.py > $O/trace.log 2>&1
python tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_trace.md 2>> $O/err.txt
rm -rf $O/trace
cat $O/kernel_trace.md | cut -c1-200; tail -2 $O/trace.log
