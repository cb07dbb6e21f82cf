#!/bin/bash
# A/B of library variants / env settings with tools/attic/repeat_bench.py (medians over fresh contexts in one process),
# interleaved ROUNDS times on the same box:  tools/attic/ab_repeat.sh "ENV=.. ENV=.." "ENV=.." ...
for r in $(seq ${ROUNDS:-2}); do
  for envs in "$@"; do
    echo "=== r$r $envs: $(env REPS=${REPS:-4} $envs python tools/attic/repeat_bench.py 2>/dev/null | tail -1)"
  done
done
