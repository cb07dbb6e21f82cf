#!/bin/bash
# the expansion with 1 / 2 / 4 / 8 work items per workgroup (ADDER_HIP_EXPAND_ITEMS): headline ms per step, frame kernel, scan + offsets + expansion
for it in 1 2 4 8; do
  for rep in 1 2; do
    r=$(ADDER_HIP_EXPAND_ITEMS=$it python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'])")
    echo "items=$it headline: $r"
  done
  ADDER_HIP_EXPAND_ITEMS=$it CONTENT=0 T=300 python tools/ablate.py | tail -1
  ADDER_HIP_EXPAND_ITEMS=$it CRF=2,7,7 TMODE=1 DTM=7650 T=300 python tools/ablate.py | tail -1
done
