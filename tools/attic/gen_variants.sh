#!/bin/bash
# the reference-default mode (Collapse, dtm 7650, AbsoluteT) for a list of env settings: GPU ms per 120-frame step
for envs in "$@"; do
  r=$(env $envs python bench.py --steps 8 --warmup 2 --frames 120 --delta-t-max 7650 --time-mode absolute_t --no-cpu-baseline --skip-roofline --no-end-to-end 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['frame_loop_ms_hip_events'], round(d['frame_loop_ms_hip_events']*1000/120,2), 'us/frame')")
  echo "=== $envs: $r"
done
