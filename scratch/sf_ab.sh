#!/bin/bash
# A/B of spectral-filter builds: scratch/sf_ab.sh name [name ...]  (scratch/bin/pwvar/lib_<name>.so), check + time each
mkdir -p gpurun_out/sf
for n in "$@"; do
  echo "== $n"
  SONAR_HIP_LIB=$PWD/scratch/bin/pwvar/lib_$n.so python scratch/sf_check.py 2>&1 | tail -6
  for i in 1 2; do SONAR_HIP_LIB=$PWD/scratch/bin/pwvar/lib_$n.so python scratch/sf_time.py 2>&1 | grep "spectral filter"; done
done 2>&1 | tee gpurun_out/sf/ab.txt
