#!/bin/bash
# fixed cost per launch against cost per frame of the headline's frame kernel: the same bench at three temporal depths
for d in 16 32 64; do
  ADDER_HIP_FRAMES_PER_LAUNCH=$d python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('depth $d', d['ms_per_step'], r['frames_per_launch'], 'frame kernel per 64 frames', r['frame_kernel_launch_us'], 'rest', r['scan_offsets_expand_us'])"
done
