#!/bin/bash
# The lean kernel's quiet-frame loop (lean_quiet / lean_step_quiet) A/B against a build without it: the headline (busy:
# must not move), static content and a crf-3 scene at delta_t_max = 255.
cd /root/repo
V=/root/repo/build/variants/libadder_hip_nolq.so
for r in 1 2; do
  for lib in "" $V; do
    echo "r$r lib=${lib##*/}"
    echo "  headline: $(ADDER_HIP_LIB=$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('step_ms', d['ms_per_step'], 'frame_kernel_us', r['frame_kernel_launch_us'], 'post_us', r['scan_offsets_expand_us'])")"
    echo "  static:   $(ADDER_HIP_LIB=$lib CONTENT=0 T=300 python tools/ablate.py 2>/dev/null | tail -1)"
    echo "  crf3:     $(ADDER_HIP_LIB=$lib CONTENT=2 T=300 CRF=2,7,7 python tools/ablate.py 2>/dev/null | tail -1)"
  done
done
