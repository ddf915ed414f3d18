#!/bin/bash
# profiling variants of libsonar_hip.so that differ only in noise_gen.hip's compile-time switches:
#   scratch/ng_build_variants.sh name "flags" [name "flags" ...]      -> scratch/bin/ngvar/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/bin/ngvar
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
build_one() {
  name=$1; flags=$2
  (cd comfyui-sonar_amd/csrc && hipcc $BASE $flags -c noise_gen.hip -o ../../scratch/bin/ngvar/noise_gen_$name.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/ngvar/lib_$name.so scratch/bin/ngvar/noise_gen_$name.o \
    $(ls comfyui-sonar_amd/build/*.o | grep -v "/noise_gen.o")
  rm -f scratch/bin/ngvar/noise_gen_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
