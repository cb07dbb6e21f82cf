#!/bin/bash
# Does co-residency of the blocked lean kernel and the expansion pay?  Walking grids sized so that both FIT on a CU
# together (LDS: 32 KB + 30 KB per workgroup, 160 KB per CU -> 5 workgroups in total), against full grids and eager serial.
cd /root/repo
export ROUNDS=${ROUNDS:-2} REPS=${REPS:-3}
tools/attic/ab_repeat.sh "A=1" \
  "ADDER_HIP_LEAN_BLOCKS_PER_CU=3 ADDER_HIP_EXPAND_BLOCKS_PER_CU=2" \
  "ADDER_HIP_LEAN_BLOCKS_PER_CU=4 ADDER_HIP_EXPAND_BLOCKS_PER_CU=1" \
  "ADDER_HIP_LEAN_BLOCKS_PER_CU=2 ADDER_HIP_EXPAND_BLOCKS_PER_CU=3" \
  "ADDER_HIP_LEAN_BLOCKS_PER_CU=3 ADDER_HIP_EXPAND_BLOCKS_PER_CU=1" \
  "ADDER_HIP_NO_GRAPH=1"
