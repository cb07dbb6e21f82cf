#!/bin/bash
# per-event-record batches (generic / bounded Collapse): walking grids and graph vs eager, default mode at 1080p
cd /root/repo
for cfg in "A=1" "ADDER_HIP_GEN_BLOCKS_PER_CU=4 ADDER_HIP_GEN_EXPAND_BLOCKS_PER_CU=1" "ADDER_HIP_GEN_BLOCKS_PER_CU=3 ADDER_HIP_GEN_EXPAND_BLOCKS_PER_CU=1" "ADDER_HIP_GEN_BLOCKS_PER_CU=3 ADDER_HIP_GEN_EXPAND_BLOCKS_PER_CU=2" "ADDER_HIP_GEN_BLOCKS_PER_CU=4 ADDER_HIP_GEN_EXPAND_BLOCKS_PER_CU=2" "ADDER_HIP_NO_GRAPH=1"; do
  echo "== $cfg"; env $cfg DTM=7650 TMODE=0 T=256 python tools/ablate.py 2>&1 | tail -1
done
echo "== wavelanes eager"; ADDER_HIP_NO_GRAPH=1 ADDER_HIP_LIB=/root/repo/build/variants/libadder_hip_wavelanes.so DTM=7650 TMODE=0 T=256 python tools/ablate.py 2>&1 | tail -1
