#!/bin/bash
# Per-kernel times of one tools/ablate.py configuration on the GPU box: tools/kstats.sh <tag> "ENV=.. ENV=.."
TAG=${1:-k}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1)); echo "=== [$i] $envs"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$i" -o b -- python "$REPO/tools/ablate.py" > "$OUT/log$i.txt" 2>&1
  tail -1 "$OUT/log$i.txt"
  f=$(find "$OUT/s$i" -name '*kernel_stats.csv' | head -1)
  cp "$f" "$OUT/kernel_stats_$i.csv"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.5:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.2f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} pct {r['Percentage']}")
PY
  rm -rf "$OUT/s$i"
done
