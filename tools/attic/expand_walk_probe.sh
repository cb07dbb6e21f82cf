#!/bin/bash
# walking expansion grids (workgroups per CU that loop over the (frame, block) items) against full grids: busy and quiet content
cd /root/repo
for r in 1 2; do
  for cfg in "A=1" "ADDER_HIP_EXPAND_BLOCKS_PER_CU=5" "ADDER_HIP_EXPAND_BLOCKS_PER_CU=10" "ADDER_HIP_EXPAND_BLOCKS_PER_CU=20"; do
    echo "r$r $cfg"
    echo "  headline: $(env $cfg python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('step_ms', d['ms_per_step'])")"
    echo "  static:   $(env $cfg CONTENT=0 T=300 python tools/ablate.py 2>/dev/null | tail -1 | cut -c80-140)"
    echo "  crf3:     $(env $cfg CONTENT=2 T=300 CRF=2,7,7 python tools/ablate.py 2>/dev/null | tail -1 | cut -c80-140)"
  done
done
