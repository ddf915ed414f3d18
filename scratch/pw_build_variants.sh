#!/bin/bash
# Builds profiling variants of libsonar_hip.so that differ only in power_fft.hip's compile-time switches:
#   scratch/pw_build_variants.sh name "flags" [name "flags" ...]      -> scratch/bin/pwvar/lib_<name>.so
# The other objects come from the product build (comfyui-sonar_amd/build/*.o): run __graft_entry__.build() first.
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/bin/pwvar
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
build_one() {
  name=$1; flags=$2
  (cd comfyui-sonar_amd/csrc && hipcc $BASE $flags -c power_fft.hip -o ../../scratch/bin/pwvar/power_fft_$name.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/pwvar/lib_$name.so scratch/bin/pwvar/power_fft_$name.o \
    comfyui-sonar_amd/build/elementwise.o comfyui-sonar_amd/build/noise_gen.o comfyui-sonar_amd/build/dwt.o comfyui-sonar_amd/build/runtime.o comfyui-sonar_amd/build/dft_direct.o comfyui-sonar_amd/build/dtcwt.o comfyui-sonar_amd/build/plan.o comfyui-sonar_amd/build/dwt_bands.o
  rm -f scratch/bin/pwvar/power_fft_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
