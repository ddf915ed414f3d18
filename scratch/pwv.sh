#!/bin/bash
# Fast profiling variants of the library with ONLY the 128 x 128 power kernels (seconds per build instead of minutes):
#   scratch/pwv.sh name "flags" [name "flags" ...]      -> scratch/bin/pwvar/lib_<name>.so
#   SRC=/path/to/csrc scratch/pwv.sh ...                  another source tree's power_fft.hip (e.g. an archived round)
# The other objects come from the product build (comfyui-sonar_amd/build/*.o): run __graft_entry__.build() first.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
SRC=${SRC:-$ROOT/comfyui-sonar_amd/csrc}
mkdir -p scratch/bin/pwvar
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DSONAR_PW_ONLY_128"
build_one() {
  name=$1; flags=$2
  (cd $SRC && hipcc $BASE $flags -c power_fft.hip -o $ROOT/scratch/bin/pwvar/power_fft_$name.o)
  objs=$(ls comfyui-sonar_amd/build/*.o | grep -v "power_fft.o\|power_buckets_\|power_any_all")
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/pwvar/lib_$name.so scratch/bin/pwvar/power_fft_$name.o $objs
  rm -f scratch/bin/pwvar/power_fft_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
