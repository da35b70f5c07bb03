#!/bin/bash
# Builds a variant of the HIP library into build_ab/<name>/ with extra compiler flags (for A/B runs on one box):
#   tools/ab_build.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab/$name
cd icp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wall -Wno-unused-result "$@" -shared -o ../../build_ab/$name/libicpx.so icpx_api.hip icpx_edt.hip icpx_nn.hip icpx_bnb.hip
cp ../libicpx_host.so ../../build_ab/$name/
