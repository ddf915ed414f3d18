#!/bin/bash
# Profiling variants of libsonar_hip.so that differ only in dwt.hip's compile-time switches -> scratch/bin/dwtvar/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/bin/dwtvar
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
build_one() {
  name=$1; flags=$2
  (cd comfyui-sonar_amd/csrc && hipcc $BASE $flags -c dwt.hip -o ../../scratch/bin/dwtvar/dwt_$name.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/dwtvar/lib_$name.so scratch/bin/dwtvar/dwt_$name.o \
    comfyui-sonar_amd/build/elementwise.o comfyui-sonar_amd/build/noise_gen.o comfyui-sonar_amd/build/power_fft.o comfyui-sonar_amd/build/runtime.o
  rm -f scratch/bin/dwtvar/dwt_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
