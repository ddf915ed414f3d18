#!/bin/bash
# profiling variants of the library that differ in dwt.hip (the tile kernels, the low-pass kernel)'s compile-time switches: scratch/bandsv.sh name "flags" ... -> scratch/bin/pwvar/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/bin/pwvar
build_one() {
  name=$1; flags=$2
  (cd comfyui-sonar_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $flags -c dwt.hip -o ../../scratch/bin/pwvar/dwt_$name.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/pwvar/lib_$name.so scratch/bin/pwvar/dwt_$name.o $(ls comfyui-sonar_amd/build/*.o | grep -v "/dwt.o")
  rm -f scratch/bin/pwvar/dwt_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
