#!/bin/bash
# fixed cost per launch against cost per frame of the default mode's frame kernel (adder_rr_kernel): three temporal depths
for tm in delta_t absolute_t; do
for d in 16 32 64; do
  ADDER_HIP_NO_GRAPH=1 ADDER_HIP_FRAMES_PER_LAUNCH=$d python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --delta-t-max 7650 --time-mode $tm 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tm depth $d', d['ms_per_step'], r['frames_per_launch'], 'frame kernel per launch', r['frame_kernel_launch_us'], 'rest', r['scan_offsets_expand_us'])"
done; done
