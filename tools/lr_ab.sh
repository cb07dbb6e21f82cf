#!/bin/bash
# A/B on ONE box: the tree's library against variants under build/variants (tools/build_variants.sh; libadder_hip_base.so =
# an earlier commit's sources).  Processes differ by +-7 % in the expansion's time (where their buffers land), so every
# variant runs REPS times, interleaved, and the medians are printed.
REPS=${REPS:-5}
VARIANTS=${VARIANTS:-"default build/variants/libadder_hip_base.so"}
: > gpurun_out/lr_ab_raw.txt
for rep in $(seq $REPS); do
for lib in $VARIANTS; do
  l=$lib; [ "$lib" = default ] && l=""
  ADDER_HIP_LIB=$l python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['ms_per_step'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'])" >> gpurun_out/lr_ab_raw.txt
done; done
python - <<'PY'
import statistics as st, collections
rows = collections.defaultdict(list)
for line in open('gpurun_out/lr_ab_raw.txt'):
    p = line.split()
    rows[p[0]].append(tuple(float(x) for x in p[1:]))
for k, v in rows.items():
    ms, fk, ex = zip(*v)
    print(f"{k:48s} step ms median {st.median(ms):.3f} min {min(ms):.3f} | frame kernel {st.median(fk):.1f} | scan+offsets+expansion median {st.median(ex):.1f} min {min(ex):.1f} max {max(ex):.1f}")
PY
