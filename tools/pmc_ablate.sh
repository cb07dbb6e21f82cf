#!/bin/bash
# PMC passes over one tools/ablate.py configuration (eager, ITERS=1): tools/pmc_ablate.sh <tag> "ENV=.. ENV=.." ["ENV=.." ...]
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains); prints the frame kernels' per-dispatch averages.
TAG=${1:-pa}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1)); echo "=== [$i] $envs"
  mkdir -p "$OUT/r$i"
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
    st=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    env $envs ITERS=1 ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/r$i/pmc_$st" -o pmc -- \
      python "$REPO/tools/ablate.py" > "$OUT/r$i/$st.log" 2>&1
  done
  python "$REPO/tools/pmc_csv_summary.py" "$OUT/r$i" > "$OUT/${TAG}_$i.csv"
  rm -rf "$OUT/r$i"
  grep -E "c[zbr]_kernel|lean|lr_kernel|rr_kernel" "$OUT/${TAG}_$i.csv"
done
