#!/bin/bash
# the tuned graph (candidate 0 = one stream, the others two branches) against eager launches on one stream
# (ADDER_HIP_NO_GRAPH=1) and the two-branch graphs alone (the round-3 behaviour: ADDER_HIP_GRAPH_SKIP_SERIAL... = base lib)
REPS=${REPS:-5}
: > gpurun_out/graph_vs_eager_raw.txt
for rep in $(seq $REPS); do
for envs in "A=1" "ADDER_HIP_NO_GRAPH=1" "ADDER_HIP_LIB=build/variants/libadder_hip_base.so"; do
  env $envs ADDER_HIP_DEBUG_TUNE=1 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline 2>gpurun_out/tune_err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$envs', d['ms_per_step'])" >> gpurun_out/graph_vs_eager_raw.txt
  grep "graph candidates" gpurun_out/tune_err.txt | head -2 >> gpurun_out/graph_vs_eager_raw.txt
done; done
python - <<'PY'
import statistics as st, collections
rows = collections.defaultdict(list)
for line in open('gpurun_out/graph_vs_eager_raw.txt'):
    if line.startswith('[adder_hip]'): print(line.strip()); continue
    p = line.split(); rows[p[0]].append(float(p[1]))
for k, v in rows.items(): print(k, 'median', st.median(v), 'min', min(v), 'max', max(v), sorted(v))
PY
