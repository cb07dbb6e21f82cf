#!/bin/bash
# Builds A/B variants of libadder_hip.so into build/variants/ (travels with gpurun; not in git):
#   tools/build_variants.sh name1 "-DFLAG=1 -DOTHER=2" name2 "..." ...
# Select one at run time with ADDER_HIP_LIB=build/variants/libadder_hip_<name>.so
# Only the two kernel files are recompiled with the flags; the other objects come from the tree's build (make first).
set -e
cd "$(dirname "$0")/../adder-codec-rs_amd"
make -j8 libadder_hip.so > /dev/null
mkdir -p ../build/variants
OTHER=$(ls obj/*.o | grep -v "adder_kernels.hip.o\|adder_lp_kernels.hip.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    for f in adder_kernels adder_lp_kernels; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags -x hip -c csrc/$f.hip -o ../build/variants/${name}_$f.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC ../build/variants/${name}_adder_kernels.o ../build/variants/${name}_adder_lp_kernels.o $OTHER -shared -o ../build/variants/libadder_hip_$name.so
    rm -f ../build/variants/${name}_adder_kernels.o ../build/variants/${name}_adder_lp_kernels.o
  ) &
done
wait
ls -la ../build/variants/
