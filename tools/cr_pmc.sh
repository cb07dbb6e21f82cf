#!/bin/bash
# Counters of the default-mode frame kernels: adder_cr_kernel (constant runs) against adder_cb_kernel (ADDER_HIP_NO_CR=1),
# both time modes: instruction counts, busy / wait cycles, LDS conflicts.  tools/cr_pmc.sh [extra env]
REPO=$(pwd); OUT=$REPO/gpurun_out/crpmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() {  # tag, env, bench args
  local tag=$1 envs=$2; shift; shift
  mkdir -p $OUT/$tag
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    st=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    env $envs $EXTRA ADDER_HIP_NO_GRAPH=1 ADDER_BENCH_PLAN_STEPS=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag/pmc_$st -o pmc -- \
      python $REPO/bench.py --steps 1 --warmup 0 --frames 128 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline --delta-t-max 7650 $* > $OUT/$tag/$st.log 2>&1
  done
  python $REPO/tools/pmc_csv_summary.py $OUT/$tag > $OUT/${tag}_pmc.csv
  rm -rf $OUT/$tag
  grep -E "cr_kernel|cb_kernel" $OUT/${tag}_pmc.csv
  # kernel durations
  env $envs $EXTRA ADDER_HIP_NO_GRAPH=1 ADDER_BENCH_PLAN_STEPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$tag -o b -- \
      python $REPO/bench.py --steps 3 --warmup 1 --frames 128 --no-cpu-baseline --no-end-to-end --no-secondary --skip-roofline --delta-t-max 7650 $* > $OUT/ks_$tag.log 2>&1
  find $OUT/ks_$tag -name '*kernel_stats.csv' -exec cp {} $OUT/${tag}_kernel_stats.csv \;
  rm -rf $OUT/ks_$tag
  grep -E "cr_kernel|cb_kernel|expand" $OUT/${tag}_kernel_stats.csv | cut -c1-160
}
EXTRA=${1:-A=1}
run cr_delta "A=1"
run cr_abs "A=1" --time-mode absolute_t
run cb_delta "ADDER_HIP_NO_CR=1"
run cb_abs "ADDER_HIP_NO_CR=1" --time-mode absolute_t
