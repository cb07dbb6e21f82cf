#!/bin/bash
# Per-DISPATCH durations (in launch order) of the big kernels of one tools/ablate.py configuration on the GPU box:
#   tools/ktrace.sh <tag> "ENV=.. ENV=.." ["ENV=.." ...]
# (kstats.sh gives the averages; this shows how a batch's launches differ -- the chunk with the pop against the quiet ones)
TAG=${1:-k}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1)); echo "=== [$i] $envs"
  env $envs rocprofv3 --kernel-trace --output-format csv -d "$OUT/s$i" -o b -- python "$REPO/tools/ablate.py" > "$OUT/log$i.txt" 2>&1
  tail -1 "$OUT/log$i.txt"
  f=$(find "$OUT/s$i" -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'synth' in n or 'rocclr' in n or 'publish' in n:
        continue
    short = n.split('(')[0].replace('void adder::', '').replace('adder::', '')
    per[short].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in per.items():
    print(f"{k:44s} n={len(v):3d} us: " + ' '.join(f"{x:.0f}" for x in v[-12:]))
PY
  rm -rf "$OUT/s$i"
done
