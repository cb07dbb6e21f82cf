#!/bin/bash
# kernel start/end timestamps of a short bench run -> do the expansion and the next chunk's frame kernel overlap?
REPO=$(pwd); OUT=$REPO/gpurun_out/ovl; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env $1 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-roofline --no-end-to-end > $OUT/log.txt 2>&1
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40],r.get('Queue_Id','?'),r.get('Stream_Id','?')) for r in rows if 'adder' in r['Kernel_Name'] and 'synth' not in r['Kernel_Name']]
ks.sort()
t0=ks[0][0]
# print the last 40 kernels
for s,e,n,q,st in ks[-44:]:
    print(f"{(s-t0)/1e3:10.1f} {(e-t0)/1e3:10.1f} dur {(e-s)/1e3:7.1f} q{q} s{st} {n}")
PY
rm -rf $OUT/t
