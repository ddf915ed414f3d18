#!/bin/bash
# Profiling variants of libsonar_hip.so that differ in ONE translation unit's compile-time switches:
#   scratch/build_variant.sh <unit, e.g. dwt> name "flags" [name "flags" ...]      -> scratch/bin/var/lib_<name>.so
# The other objects come from the product build (comfyui-sonar_amd/build/*.o): run __graft_entry__.build() first.
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/bin/var
unit=$1; shift
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
# EXCLUDE="a b": objects left out of the link (power_fft with -DSONAR_PW_ONLY_128: EXCLUDE="power_any_all power_buckets_a power_buckets_b")
others=$(ls comfyui-sonar_amd/build/*.o | grep -v "/$unit.o")
for x in $EXCLUDE; do others=$(echo "$others" | grep -v "/$x.o"); done
build_one() {
  name=$1; flags=$2
  (cd comfyui-sonar_amd/csrc && hipcc $BASE $flags -c $unit.hip -o ../../scratch/bin/var/${unit}_$name.o 2>/dev/null)
  hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/var/lib_$name.so scratch/bin/var/${unit}_$name.o $others
  rm -f scratch/bin/var/${unit}_$name.o
  echo built $name
}
while [ $# -gt 1 ]; do build_one "$1" "$2" & shift 2; done
wait
