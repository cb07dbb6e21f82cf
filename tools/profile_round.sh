#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# 1. kernel trace + stats of the default bench command,
# 2. the same bench without its per-launch timing passes (every frame-kernel launch is a default-depth one),
# 3. separate --pmc passes (never combined with other trace domains) of a 160-frame eager run at the default
#    temporal depth, and of a 96-frame run with ONE frame per launch (the adder_lean1_kernel rows),
# then writes summaries under gpurun_out/profiles_<round>/ (copy them into profiles/).
set -u
ROUND=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --secondary-ms 30 > "$OUT/bench_under_rocprof.log" 2>&1
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_kernel_stats.csv" \;
find "$OUT/stats" -name '*domain_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_domain_stats.csv" \;

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary > "$OUT/bench_default_depth_under_rocprof.log" 2>&1
find "$OUT/stats2" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_default_depth_kernel_stats.csv" \;

# 2b. the same, eager on ONE stream (ADDER_HIP_NO_GRAPH=1): the kernels run one after the other at full grids, which is
#     what bench.py's HIP-event pairs time for its `roofline` block (in the default run the frame kernel and the
#     expansion share the chip, so their trace durations overlap and are longer)
ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary > "$OUT/bench_eager_under_rocprof.log" 2>&1
find "$OUT/stats3" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_bench_eager_serial_kernel_stats.csv" \;

pmc_passes() {  # $1 = tag, $2 = extra env, $3.. = bench args
    local tag=$1 envs=$2; shift; shift
    local cmd="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline $*"
    mkdir -p "$OUT/$tag"
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
               "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
        local st=$(echo "$set" | tr ' ' '_' | cut -c1-40)
        env $envs ADDER_HIP_NO_GRAPH=1 ADDER_BENCH_PLAN_STEPS=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$tag/pmc_$st" -o pmc -- \
            $cmd > "$OUT/$tag/pmc_$st.log" 2>&1
    done
    # (FPL: frames every frame-kernel launch of the command covered -- bench.py scales the traffic by it; the frame counts
    #  below are multiples of the 64-frame chunk, so that all launches are alike)
    python "$REPO/tools/pmc_csv_summary.py" "$OUT/$tag" --traffic "$OUT/${ROUND}_traffic_$tag.json" --frames-per-launch ${FPL:-64} > "$OUT/${ROUND}_pmc_$tag.csv"
    rm -rf "$OUT/$tag"/pmc_*/
}
pmc_passes default "A=1" --frames 128
FPL=1 pmc_passes one_frame_per_launch "ADDER_HIP_FRAMES_PER_LAUNCH=1" --frames 96 --output events
pmc_passes default_mode_dtm7650_abs "A=1" --frames 128 --delta-t-max 7650 --time-mode absolute_t --output events
# 4. the reference default mode (Collapse, AbsoluteT, delta_t_max 7650: adder_cb_kernel) eager on one stream: kernel stats
ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats4" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary --delta-t-max 7650 --time-mode absolute_t --output events > "$OUT/bench_default_mode_under_rocprof.log" 2>&1
find "$OUT/stats4" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_default_mode_eager_kernel_stats.csv" \;
# (every row below runs with --output events: AdderEvents as the step's output, as bench.py's secondary legs run these modes;
#  the default command of 1.-3. writes the raw sink's records)
# 5. (round 4) the kernels VERDICT r3 found without evidence: adder_cb_kernel<false> (DeltaT default mode), adder_frame_kernel
#    (Normal mode), and the one-frame-per-launch pipeline's scan / expansion -- PMC passes + eager kernel stats each
pmc_passes default_mode_dtm7650_delta "A=1" --frames 128 --delta-t-max 7650 --output events
pmc_passes normal_dtm255_delta "A=1" --frames 128 --multi-mode normal --output events
kstats() {  # $1 = tag, $2 = extra env, $3.. = bench args
    local tag=$1 envs=$2; shift; shift
    env $envs ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_$tag" -o bench -- \
        python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary $* > "$OUT/bench_${tag}_under_rocprof.log" 2>&1
    find "$OUT/ks_$tag" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_${tag}_eager_kernel_stats.csv" \;
    rm -rf "$OUT/ks_$tag"
}
# (round 4: at crf 0 the default mode runs adder_rr_kernel; ADDER_HIP_NO_RR=1 gives adder_cr_kernel's rows,
#  ADDER_HIP_NO_RR=1 ADDER_HIP_NO_CR=1 the bounded Collapse kernel's)
pmc_passes cr_dtm7650_delta "ADDER_HIP_NO_RR=1" --frames 128 --delta-t-max 7650 --output events
pmc_passes cr_dtm7650_abs "ADDER_HIP_NO_RR=1" --frames 128 --delta-t-max 7650 --time-mode absolute_t --output events
pmc_passes cb_dtm7650_delta "ADDER_HIP_NO_RR=1 ADDER_HIP_NO_CR=1" --frames 128 --delta-t-max 7650 --output events
pmc_passes cb_dtm7650_abs "ADDER_HIP_NO_RR=1 ADDER_HIP_NO_CR=1" --frames 128 --delta-t-max 7650 --time-mode absolute_t --output events
pmc_passes lean_step_no_runs "ADDER_HIP_NO_LR=1" --frames 128 --output events
pmc_passes events_output "A=1" --frames 128 --output events
kstats cr_dtm7650_delta "ADDER_HIP_NO_RR=1" --delta-t-max 7650 --output events
kstats cr_dtm7650_abs "ADDER_HIP_NO_RR=1" --delta-t-max 7650 --time-mode absolute_t --output events
kstats cb_dtm7650_delta "ADDER_HIP_NO_RR=1 ADDER_HIP_NO_CR=1" --delta-t-max 7650 --output events
kstats cb_dtm7650_abs "ADDER_HIP_NO_RR=1 ADDER_HIP_NO_CR=1" --delta-t-max 7650 --time-mode absolute_t --output events
kstats events_output "A=1" --output events
kstats default_mode_delta "A=1" --delta-t-max 7650 --output events
kstats normal_dtm255_delta "A=1" --multi-mode normal --output events
kstats normal_dtm7650_abs "A=1" --multi-mode normal --delta-t-max 7650 --time-mode absolute_t --output events
kstats one_frame_per_launch "ADDER_HIP_FRAMES_PER_LAUNCH=1" --frames 128 --output events
# 6. (round 5) the QUIET legs -- static content through the lean-runs kernel's quiet groups, the reference's default mode at
#    its default quality (crf-3 numbers) through the bounded Collapse kernel's, config 5's shape (4K RGB): counters + kernel stats
pmc_passes quiet_static "A=1" --frames 128 --content static --output events
pmc_passes quiet_default_quality "A=1" --frames 128 --delta-t-max 7650 --time-mode absolute_t --crf-numbers 2,7,7 --output events
kstats quiet_static "A=1" --content static --output events
kstats quiet_default_quality "A=1" --delta-t-max 7650 --time-mode absolute_t --crf-numbers 2,7,7 --output events
kstats quiet_config5_shape "A=1" --width 3840 --height 2160 --channels 3 --frames 64 --delta-t-max 7650 --time-mode absolute_t --crf-numbers 2,7,7 --output events
rm -rf "$OUT"/stats "$OUT"/stats2 "$OUT"/stats3 "$OUT"/stats4
ls -la "$OUT"
