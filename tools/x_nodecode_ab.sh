#!/bin/bash
# the expansion with its stores only (-DADDER_DBG_X_NODECODE=1, tools/build_variants.sh nodecode ...) against the full one
for lib in "" build/variants/libadder_hip_nodecode.so "" build/variants/libadder_hip_nodecode.so; do
  ADDER_HIP_LIB=$lib python bench.py --steps 32 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-default}', d['ms_per_step'], d['value'], r['frac'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'])"
done
