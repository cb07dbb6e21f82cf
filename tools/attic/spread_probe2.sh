#!/bin/bash
# context lottery: scan + offsets + expansion per chunk (eager pass of bench.py) over fresh processes, slot layout vs logs
cd /root/repo
for r in 1 2 3 4 5 6; do
  for cfg in "A=1" "ADDER_HIP_LEAN_LOG=1"; do
    echo "r$r $cfg: $(env $cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('step_ms', d['ms_per_step'], 'lean_us', r['frame_kernel_launch_us'], 'post_us', r['scan_offsets_expand_us'])")"
  done
done
