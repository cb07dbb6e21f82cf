#!/bin/bash
# compiler-flag variants of the whole library: headline kernels (eager pass of bench.py) + the default mode at crf 0 / 3
cd /root/repo
V=/root/repo/build/variants
for r in 1 2 3; do
  for lib in "" $(ls $V/*.so); do
    echo "r$r lib=${lib##*/}: $(ADDER_HIP_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['roofline_one_frame_per_launch']
print('step_ms', d['ms_per_step'], 'lean_us', r['frame_kernel_launch_us'], 'post_us', r['scan_offsets_expand_us'], 'one_frame_us', o['launch_avg_us'])")"
  done
done
