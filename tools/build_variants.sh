#!/bin/bash
# Builds A/B variants of libadder_hip.so into build/variants/ (travels with gpurun; not in git):
#   tools/build_variants.sh name1 "-DFLAG=1 -DOTHER=2" name2 "..." ...
# Select one at run time with ADDER_HIP_LIB=build/variants/libadder_hip_<name>.so
set -e
cd "$(dirname "$0")/../adder-codec-rs_amd"
mkdir -p ../build/variants
SRCS="csrc/adder_kernels.hip csrc/adder_hip_api.cpp csrc/adder_raw_sink.cpp csrc/adder_framer_kernels.hip csrc/adder_framer_api.cpp csrc/adder_compressed.cpp csrc/adder_sparse.hip"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags -x hip $SRCS -shared -o ../build/variants/libadder_hip_$name.so &
done
wait
ls -la ../build/variants/
