#!/bin/bash
# AbsoluteT lean kernel with a 96-VGPR cap (5 waves per SIMD instead of 4): headline shape in both time modes
cd /root/repo
V=/root/repo/build/variants
for r in 1 2 3; do
  for lib in "" "$V/libadder_hip_v96.so"; do
    for tm in 1 0; do
      echo "r$r lib=${lib##*/} tmode=$tm: $(ADDER_HIP_LIB=$lib TMODE=$tm T=300 python tools/ablate.py 2>/dev/null | tail -1)"
    done
  done
done
