#!/bin/bash
# Instruction counts per kernel (one rocprofv3 --pmc pass) for a list of env sets: tools/pmc_insts.sh <tag> "ENV=.." "ENV=.." ...
TAG=${1:-pi}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1)); echo "=== [$i] $envs"
  env $envs ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVES --output-format csv -d "$OUT/pmc_$i" -o pmc -- \
      python $REPO/bench.py --steps 1 --warmup 0 --frames 128 --no-cpu-baseline --skip-roofline --no-secondary --no-end-to-end > "$OUT/log$i.txt" 2>&1
  mkdir -p "$OUT/p$i"; mv "$OUT/pmc_$i" "$OUT/p$i/pmc_x"
  python "$REPO/tools/pmc_csv_summary.py" "$OUT/p$i" | grep -E "lp_kernel|lpx|expand" | awk -F, '{printf "%-40s %-18s %12.0f\n", $1, $2, $4}'
  rm -rf "$OUT/p$i"
done
