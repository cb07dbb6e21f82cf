#!/bin/bash
# ms_per_step of the default bench for a list of env settings: tools/bench_variants.sh "ENV=.." "ENV=.." ...
for envs in "$@"; do
  r=$(env $envs python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline --skip-roofline --no-end-to-end 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "=== $envs: $r"
done
