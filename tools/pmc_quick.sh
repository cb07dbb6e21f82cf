#!/bin/bash
# PMC passes of a short eager bench run on the GPU box: tools/pmc_quick.sh <tag> "<ENV=.. ENV=..>" [bench args...]
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains); prints per-kernel averages.
TAG=${1:-pmc}; ENVS=${2:-A=1}; shift; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 1 --warmup 0 --frames 96 --no-cpu-baseline --skip-roofline $*"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    env $ENVS ADDER_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        $CMD > "$OUT/pmc_$tag.log" 2>&1
done
python "$REPO/tools/pmc_csv_summary.py" "$OUT" > "$OUT/${TAG}_pmc_summary.csv"
rm -rf "$OUT"/pmc_*/
grep -E "lean|expand|frame_kernel|scan|lp_kernel|lr_kernel" "$OUT/${TAG}_pmc_summary.csv"
