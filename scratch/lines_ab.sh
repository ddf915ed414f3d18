#!/bin/bash
# A/B of library builds on the planes beyond LDS: scratch/lines_ab.sh lib_a.so lib_b.so ...
for l in "$@"; do
  echo "== $l"
  SONAR_HIP_LIB=$PWD/$l python scratch/size_sweep.py 256x256 2>&1 | grep -v amdgpu
  SONAR_HIP_LIB=$PWD/$l python -m pytest tests/test_gpu_kernels.py -q -k "beyond_lds or block" 2>&1 | tail -1
done
