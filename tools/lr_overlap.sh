#!/bin/bash
# does the lean-runs frame kernel (scalar-bound, 46 VGPRs, 16 KB of LDS per workgroup) co-run with the expansion?
# walking grids (workgroups per CU) in the captured two-stream graph against full grids
for envs in "A=1" "ADDER_HIP_LEAN_BLOCKS_PER_CU=3 ADDER_HIP_EXPAND_BLOCKS_PER_CU=2" "ADDER_HIP_LEAN_BLOCKS_PER_CU=4 ADDER_HIP_EXPAND_BLOCKS_PER_CU=2" \
            "ADDER_HIP_LEAN_BLOCKS_PER_CU=4 ADDER_HIP_EXPAND_BLOCKS_PER_CU=3" "ADDER_HIP_LEAN_BLOCKS_PER_CU=5 ADDER_HIP_EXPAND_BLOCKS_PER_CU=3" \
            "ADDER_HIP_LEAN_BLOCKS_PER_CU=3 ADDER_HIP_EXPAND_BLOCKS_PER_CU=3" "ADDER_HIP_LEAN_BLOCKS_PER_CU=2 ADDER_HIP_EXPAND_BLOCKS_PER_CU=3" \
            "ADDER_HIP_LEAN_BLOCKS_PER_CU=6 ADDER_HIP_EXPAND_BLOCKS_PER_CU=2" "ADDER_HIP_NO_GRAPH=1" "A=2"; do
  for rep in 1 2; do
  r=$(env $envs python bench.py --steps 32 --warmup 3 --no-cpu-baseline --skip-roofline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$envs: $r"
  done
done
