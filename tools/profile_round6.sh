#!/bin/bash
# Round 6's rocprofv3 evidence (run through gpurun from the repo root): tools/profile_round6.sh [r06]
# The trimmed form of tools/profile_round.sh: kernel stats of the default bench command (tuned graph, and eager on one stream --
# what bench.py's HIP-event pairs time), separate --pmc passes (never combined with other trace domains) for the default
# command, its AdderEvents form and the one-frame-per-launch regime, kernel stats of the quiet legs and of the default mode.
set -u
ROUND=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
kstats() {  # $1 = tag, $2 = extra env, $3.. = bench args
    local tag=$1 envs=$2; shift; shift
    env $envs rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_$tag" -o bench -- \
        python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary $* > "$OUT/bench_${tag}_under_rocprof.log" 2>&1
    find "$OUT/ks_$tag" -name '*kernel_stats.csv' -exec cp {} "$OUT/${ROUND}_${tag}_kernel_stats.csv" \;
    rm -rf "$OUT/ks_$tag"
}
pmc_passes() {  # $1 = tag, $2 = extra env, $3.. = bench args
    local tag=$1 envs=$2; shift; shift
    local cmd="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline $*"
    mkdir -p "$OUT/$tag"
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
               "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
        local st=$(echo "$set" | tr ' ' '_' | cut -c1-40)
        env $envs ADDER_HIP_NO_GRAPH=1 ADDER_BENCH_PLAN_STEPS=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$tag/pmc_$st" -o pmc -- \
            $cmd > "$OUT/$tag/pmc_$st.log" 2>&1
    done
    python "$REPO/tools/pmc_csv_summary.py" "$OUT/$tag" --traffic "$OUT/${ROUND}_traffic_$tag.json" --frames-per-launch ${FPL:-64} > "$OUT/${ROUND}_pmc_$tag.csv"
    rm -rf "$OUT/$tag"
}
kstats bench "A=1"
kstats bench_eager_serial "ADDER_HIP_NO_GRAPH=1"
pmc_passes default "A=1" --frames 128
pmc_passes events_output "A=1" --frames 128 --output events
FPL=1 pmc_passes one_frame_per_launch "ADDER_HIP_FRAMES_PER_LAUNCH=1" --frames 96 --output events
kstats one_frame_per_launch_eager "ADDER_HIP_NO_GRAPH=1 ADDER_HIP_FRAMES_PER_LAUNCH=1" --output events
kstats lr_no_lp_eager "ADDER_HIP_NO_GRAPH=1 ADDER_HIP_NO_LP=1"
kstats quiet_static_eager "ADDER_HIP_NO_GRAPH=1" --content static
kstats quiet_default_quality_eager "ADDER_HIP_NO_GRAPH=1" --delta-t-max 7650 --time-mode absolute_t --crf-numbers 2,7,7 --output events
kstats default_mode_eager "ADDER_HIP_NO_GRAPH=1" --delta-t-max 7650 --time-mode absolute_t --output events
kstats c3_rgb_eager "ADDER_HIP_NO_GRAPH=1" --channels 3
ls "$OUT"
