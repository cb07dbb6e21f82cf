#!/bin/bash
# PMC passes of one tools/ablate.py configuration (eager, one stream): tools/attic/pmc_ablate.sh <tag> "ENV=.. ENV=.." [sets]
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains); prints per-kernel averages.
TAG=${1:-pmc}; ENVS=${2:-A=1}; MODE=${3:-inst}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
if [ "$MODE" = "inst" ]; then
  SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU")
else
  SETS=("FETCH_SIZE" "WRITE_SIZE")
fi
for set in "${SETS[@]}"; do
    tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    env $ENVS ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        python $REPO/tools/ablate.py > "$OUT/pmc_$tag.log" 2>&1
done
python "$REPO/tools/pmc_csv_summary.py" "$OUT" > "$OUT/${TAG}_pmc_summary.csv"
rm -rf "$OUT"/pmc_*/
grep -E "lean|expand|frame_kernel|cb_kernel" "$OUT/${TAG}_pmc_summary.csv"
