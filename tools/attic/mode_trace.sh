#!/bin/bash
# kernel trace of several contexts in one process: which queues do the two branches use, and do they overlap?
REPO=$(pwd); OUT=$REPO/gpurun_out/mode; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env $1 REPS=${REPS:-6} rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python $REPO/tools/attic/repeat_bench.py > $OUT/log.txt 2>&1
grep "\[(" $OUT/log.txt | cut -c1-400
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id','?')) for r in rows]
ks.sort()
# split into contexts at synth kernels
ctx=[];cur=[]
for k in ks:
    if 'synth' in k[2]:
        if cur: ctx.append(cur)
        cur=[]
    elif 'adder::' in k[2]: cur.append(k)
if cur: ctx.append(cur)
for i,c in enumerate(ctx):
    lean=[k for k in c if 'lean_kernel' in k[2]]; exp=[k for k in c if 'expand_kernel' in k[2]]
    ql=collections.Counter(k[3] for k in lean); qe=collections.Counter(k[3] for k in exp)
    # overlap time between lean and expand intervals (last 40% of kernels)
    L=lean[len(lean)//2:]; E=exp[len(exp)//2:]
    ov=0
    for a in L:
        for b in E:
            ov+=max(0,min(a[1],b[1])-max(a[0],b[0]))
    tl=sum(a[1]-a[0] for a in L); te=sum(b[1]-b[0] for b in E)
    print(f"ctx {i}: lean queues {dict(ql)} expand queues {dict(qe)} lean_avg_us {tl/len(L)/1e3:.1f} exp_avg_us {te/len(E)/1e3:.1f} overlap_frac_of_expand {ov/te:.2f}")
PY
rm -rf $OUT/t
