#!/bin/bash
# eager kernel stats (+ instruction counters) of the default mode at crf 0 through adder_rr_kernel, both time modes
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/rr_prof; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for tm in delta_t absolute_t; do
  ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_$tm" -o bench -- \
    python "$REPO/bench.py" --steps 5 --warmup 2 --skip-roofline --no-cpu-baseline --no-end-to-end --no-secondary --delta-t-max 7650 --time-mode $tm > "$OUT/bench_$tm.log" 2>&1
  find "$OUT/ks_$tm" -name '*kernel_stats.csv' -exec cp {} "$OUT/rr_${tm}_kernel_stats.csv" \;
  rm -rf "$OUT/ks_$tm"
  head -8 "$OUT/rr_${tm}_kernel_stats.csv" | cut -c1-200
done
if [ "${1:-}" = "pmc" ]; then
  for tm in delta_t absolute_t; do
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
      st=$(echo "$set" | tr ' ' '_' | cut -c1-40)
      mkdir -p "$OUT/$tm"
      ADDER_HIP_NO_GRAPH=1 ADDER_BENCH_PLAN_STEPS=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$tm/pmc_$st" -o pmc -- \
        python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline --frames 128 --delta-t-max 7650 --time-mode $tm > "$OUT/$tm/pmc_$st.log" 2>&1
    done
    python "$REPO/tools/pmc_csv_summary.py" "$OUT/$tm" --traffic "$OUT/rr_traffic_$tm.json" > "$OUT/rr_pmc_$tm.csv"
    rm -rf "$OUT/$tm"/pmc_*/
    cat "$OUT/rr_pmc_$tm.csv" | cut -c1-250
  done
fi
