#!/bin/bash
# usage: scratch/build_fftbench.sh <tag> [-D flags...]
tag=$1; shift
cd /root/repo/comfyui-sonar_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-pass-failed "$@" power_fft.hip runtime.hip ../../scratch/fftbench.cpp -o ../../scratch/bin/fftbench_$tag 2>&1 | grep -E "error" -A5
