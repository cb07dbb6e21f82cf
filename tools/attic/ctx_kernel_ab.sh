#!/bin/bash
# Which kernel differs between a fast and a slow context?  tools/attic/ctx_spread_probe.py under rocprofv3: kernel trace (or, with
# PMC="counter ...", one counter pass), mean per kernel and per context (contexts run one after the other: dispatch order).
REPO=$(pwd); OUT=$REPO/gpurun_out/ctxab; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
K=${K:-8}
if [ -n "$PMC" ]; then
  K=$K ROUNDS=1 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/t -o k -- python $REPO/tools/attic/ctx_spread_probe.py > $OUT/log.txt 2>&1
else
  K=$K ROUNDS=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python $REPO/tools/attic/ctx_spread_probe.py > $OUT/log.txt 2>&1
fi
grep "round" $OUT/log.txt | cut -c1-200
python3 - "$OUT/t" "$K" <<'PY'
import csv,sys,glob,collections
d,K=sys.argv[1],int(sys.argv[2])
tr=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(tr)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
cc=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
for name in ('lean_kernel','expand_kernel'):
    ks=[r for r in rows if name in r['Kernel_Name']]
    per=len(ks)//K
    print(name,'dispatches',len(ks),'per ctx',per)
    print('  mean us per ctx:',[round(sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in ks[i*per:(i+1)*per])/per/1e3,1) for i in range(K)])
if cc:
    crow=list(csv.DictReader(open(cc[0])))
    by=collections.defaultdict(list)
    for r in crow:
        by[(r['Counter_Name'],)].append(r)
    for (cn,),rs in by.items():
        rs.sort(key=lambda r:int(r.get('Dispatch_Id',0)))
        for name in ('lean_kernel','expand_kernel'):
            ks=[r for r in rs if name in r['Kernel_Name']]
            per=len(ks)//K
            if per: print(cn,name,[round(sum(float(r['Counter_Value']) for r in ks[i*per:(i+1)*per])/per,1) for i in range(K)])
PY
rm -rf $OUT/t
