#!/bin/bash
# CU-masked streams (ADDER_HIP_CU_SPLIT=n: the frame kernel of chunk k+1 on n CUs, the expansion of chunk k on the others)
# against the graph, eager one-stream and eager two-stream submissions: ms per headline step, a parity run under the
# split, and the kernel timeline of the best split.  tools/cu_split_sweep.sh [n ...]  -> gpurun_out/cusplit/
REPO=$(pwd); OUT=$REPO/gpurun_out/cusplit; mkdir -p $OUT
SPLITS=${@:-"128 144 160 168 176 184 192 208 224"}
run() {  # env settings -> "ms_per_step value"
  env $1 python bench.py --steps ${STEPS:-32} --warmup 3 --no-cpu-baseline --skip-roofline --no-end-to-end --no-secondary 2>$OUT/err.txt | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['events_per_pixel_frame'])" || tail -5 $OUT/err.txt
}
echo "setting,ms_per_step,mpixels_per_s,events_per_pixel_frame" > $OUT/sweep.csv
for rep in 1 2; do
  for envs in "ADDER_X=graph" "ADDER_HIP_NO_GRAPH=1" "ADDER_HIP_NO_GRAPH=2"; do
    r=$(run "$envs"); echo "$envs,$(echo $r | tr ' ' ',')" | tee -a $OUT/sweep.csv
  done
  for n in $SPLITS; do
    r=$(run "ADDER_HIP_CU_SPLIT=$n"); echo "ADDER_HIP_CU_SPLIT=$n,$(echo $r | tr ' ' ',')" | tee -a $OUT/sweep.csv
  done
done
best=$(grep CU_SPLIT $OUT/sweep.csv | sort -t, -k2 -n | head -1 | cut -d, -f1)
echo "best: $best" | tee -a $OUT/sweep.csv
# parity under the best split: the full-size cases that cross chunk boundaries + the lean fuzz slice
env $best python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu -k "chunk or full_size or config or fuzz or 1080" 2>&1 | tail -3 | tee $OUT/parity.txt
# kernel timeline of the best split (do chunk k+1's frame kernel and chunk k's expansion run at the same time?)
export TMPDIR=/tmp; cd /tmp
env $best rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-roofline --no-end-to-end --no-secondary > $OUT/trace_log.txt 2>&1
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$best" > $OUT/timeline.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:44],r.get('Queue_Id','?')) for r in rows if 'adder' in r['Kernel_Name'] and 'synth' not in r['Kernel_Name']]
ks.sort()
print("#", sys.argv[2], "-- last 26 kernels of the run (us from the first shown)")
ks=ks[-26:]; t0=ks[0][0]
for s,e,n,q in ks:
    print(f"{(s-t0)/1e3:10.1f} {(e-t0)/1e3:10.1f} dur {(e-s)/1e3:7.1f} q{q} {n}")
PY
rm -rf $OUT/t; cat $OUT/timeline.txt
