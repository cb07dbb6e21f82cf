#!/bin/bash
# Per-kernel A/B: the eager single-stream timing pass of bench.py's roofline block (HIP-event pairs around the frame
# kernel launches and around scan + offsets + expansion of each chunk) for a list of env settings.
for r in $(seq ${ROUNDS:-2}); do
  for envs in "$@"; do
    r_=$(env $envs python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['roofline_one_frame_per_launch']
print('step_ms', d['ms_per_step'], 'frame_kernel_us', r['frame_kernel_launch_us'], 'scan_offsets_expand_us', r['scan_offsets_expand_us'], 'one_frame_us', o['launch_avg_us'])")
    echo "=== r$r $envs: $r_"
  done
done
