#!/bin/bash
# The per-frame ring on a timeline: kernel trace (the memory-copy trace crashes rocprofv3 on this program) of tools/ring_probe.py (DQ=1: default quality, wire records),
# then per steady-state frame the start / end of the upload, the frame kernel and the hand-over chain relative to the frame's
# upload start.  tools/probes/ring_timeline.sh [tag]
TAG=${1:-ringtl}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
DQ=1 T=${T:-160} rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o r -- python "$REPO/tools/ring_probe.py" > "$OUT/log.txt" 2>&1
tail -1 "$OUT/log.txt"
python3 - "$OUT/tr" <<'PY'
import csv, glob, sys, os
root = sys.argv[1]
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    if "adder" not in n or "synth" in n: continue
    short = n.split("(")[0].replace("void adder::", "").replace("adder::", "").split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
k1 = [e for e in ev if e[2] in ("adder_cb_kernel", "adder_lean1w_kernel", "adder_lean1_kernel", "adder_lean_kernel", "adder_rr_kernel", "adder_cr_kernel")]
print("frame kernels", len(k1))
sel = k1[-45:-5]
per = [(sel[i + 1][0] - sel[i][0]) / 1e3 for i in range(len(sel) - 1)]
print("frame kernel start spacing us: median %.1f min %.1f max %.1f" % (sorted(per)[len(per) // 2], min(per), max(per)))
import collections
dur = collections.defaultdict(list)
t0, t1 = sel[0][0], sel[-1][1]
for s, e, n in ev:
    if t0 <= s <= t1: dur[n].append((e - s) / 1e3)
for n, v in dur.items():
    v.sort(); print(f"{n:32s} n={len(v):4d} median {v[len(v)//2]:7.1f} us  sum/frame {sum(v)/len(per):7.1f}")
a, bnd = sel[20][0], sel[23][0]
for s, e, n in ev:
    if a <= s < bnd: print(f"  {(s - a) / 1e3:8.1f} .. {(e - a) / 1e3:8.1f}  {n}")
PY
rm -rf "$OUT/tr"
