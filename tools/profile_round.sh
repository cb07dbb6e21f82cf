#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1. kernel trace + stats of the default bench command,
# 2. separate --pmc passes (never combined with other trace domains) of a 64-frame eager run,
# then writes summaries under gpurun_out/profiles_<round>/ (copy them into profiles/).
set -u
ROUND=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 > "$OUT/bench_under_rocprof.log" 2>&1
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_kernel_stats.csv" \;
find "$OUT/stats" -name '*domain_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_domain_stats.csv" \;

# the same bench without its per-launch timing passes: every frame-kernel launch is a default-depth one,
# so the average duration here is directly comparable with bench.py's roofline.launch_avg_us
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline > "$OUT/bench_default_depth_under_rocprof.log" 2>&1
find "$OUT/stats2" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_default_depth_kernel_stats.csv" \;

PMC_CMD="python $REPO/bench.py --steps 1 --warmup 0 --frames 160 --no-cpu-baseline --skip-roofline"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        $PMC_CMD > "$OUT/pmc_$tag.log" 2>&1
done
python "$REPO/tools/pmc_csv_summary.py" "$OUT" > "$OUT/${ROUND}_pmc_summary.csv"
python "$REPO/tools/pmc_csv_summary.py" "$OUT" --traffic "$OUT/traffic_latest.json" > /dev/null
# drop the bulky raw traces, keep logs + summaries
rm -rf "$OUT"/stats "$OUT"/stats2 "$OUT"/pmc_*/
ls -la "$OUT"
