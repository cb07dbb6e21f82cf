#!/bin/bash
# the headline with the raw sink's records as the expansion's output against AdderEvents, REPS processes each
REPS=${REPS:-4}
for rep in $(seq $REPS); do
for o in events wire; do
  python bench.py --output $o --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$o', d['ms_per_step'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'], r['frac'], (d.get('output_check') or {}).get('events_output_ms_per_step'))"
done; done
