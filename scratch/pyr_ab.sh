#!/bin/bash
# round-robin A/B of libsonar_hip.so variants on the pyramid rows (scratch/pyr_time.py): scratch/pyr_ab.sh name ... (scratch/bin/ngvar/lib_<name>.so; "head" = the product)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = head ]; then lib=comfyui-sonar_amd/libsonar_hip.so; else lib=scratch/bin/ngvar/lib_$v.so; fi
    echo "== $v (pass $rep)"; SONAR_HIP_LIB=$PWD/$lib python scratch/pyr_time.py 2>&1 | grep "us per call"
  done
done
