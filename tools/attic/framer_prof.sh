#!/bin/bash
# Kernel stats + PMC passes of tools/framer_bench.py on the GPU box: tools/attic/framer_prof.sh <tag> ["ENV=.. ENV=.."]
TAG=${1:-framer}; ENVS=${2:-A=1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
env $ENVS rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- python "$REPO/tools/framer_bench.py" > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/${TAG}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'framer' in r['Name'] or float(r['Percentage']) > 3:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.2f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f}")
PY
tail -1 "$OUT/stats.log"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
    tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    env $ENVS rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        python "$REPO/tools/framer_bench.py" > "$OUT/pmc_$tag.log" 2>&1
done
python "$REPO/tools/pmc_csv_summary.py" "$OUT" > "$OUT/${TAG}_pmc_summary.csv"
rm -rf "$OUT"/pmc_*/ "$OUT/stats"
grep -E "framer" "$OUT/${TAG}_pmc_summary.csv"
