#!/bin/bash
# the per-frame ring with wire records: the expansion storing them into the page-locked slot (default) against the
# serialising hand-over kernel (ADDER_HIP_RING_DIRECT_WIRE=0; =1: always; default: while the frames are sparse)
for rep in 1 2 3; do
for envs in "A=1" "ADDER_HIP_LIB=build/variants/libadder_hip_base.so"; do
  env $envs python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --skip-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['end_to_end']
print('$envs', 'ring events', e['per_frame_ring'].get('us_per_frame_sustained'), 'ring wire', e['per_frame_ring_wire_records'].get('value'), e['per_frame_ring_wire_records'].get('error'), 'default quality', e['default_quality_raw'].get('us_per_frame_sustained'))"
done; done
