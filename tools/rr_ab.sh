#!/bin/bash
# the default mode at crf 0 (adder_rr_kernel), both time modes: the tree's library against build/variants/libadder_hip_base.so
for rep in 1 2 3; do
for lib in "" build/variants/libadder_hip_base.so; do
for tm in delta_t absolute_t; do
  ADDER_HIP_NO_GRAPH=1 ADDER_HIP_LIB=$lib python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --delta-t-max 7650 --time-mode $tm 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-default}', '$tm', d['ms_per_step'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'])"
done; done; done
