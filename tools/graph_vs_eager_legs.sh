#!/bin/bash
# the secondary legs of the bench line under the tuned graph against eager launches on one stream (ADDER_HIP_NO_GRAPH=1)
for envs in "A=1" "ADDER_HIP_NO_GRAPH=1"; do
  env $envs python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-end-to-end --skip-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$envs', 'headline', d['ms_per_step'])
for l in d['secondary']: print('$envs', l['workload'][:64], l.get('us_per_frame'), l.get('frac'))"
done
