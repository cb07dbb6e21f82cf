#!/bin/bash
# Quick A/B of kernel times on the GPU box: tools/prof_quick.sh <tag> ["ENV=.. ENV=.." ...]
# For every env set: rocprofv3 --kernel-trace --stats of a short bench run; prints the per-kernel table.
TAG=${1:-q}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "=== [$i] $envs"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$i" -o b -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --skip-roofline $BENCH_ARGS > "$OUT/log$i.txt" 2>&1
  f=$(find "$OUT/s$i" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.2f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} pct {r['Percentage']}")
PY
  grep -o '"ms_per_step": [0-9.]*' "$OUT/log$i.txt"
  rm -rf "$OUT/s$i"
done
