#!/bin/bash
# where the framer's ingest call spends its time: HIP-event time of the call vs the kernels' own durations
# (rocprofv3 kernel trace of the same run).  tools/attic/framer_gap.sh "T=64" "T=256 DTM=7650" ...
REPO=$(pwd); OUT=$REPO/gpurun_out/fgap; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for envs in "$@"; do
  echo "=== $envs"
  env $envs python $REPO/tools/framer_bench.py 2>/dev/null | tail -1
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o b -- python $REPO/tools/framer_bench.py > $OUT/log.txt 2>&1
  tail -1 $OUT/log.txt
  f=$(find $OUT/s -name '*kernel_stats.csv' | head -1)
  grep -E "tiles_kernel|slices_kernel" $f | awk -F, '{print $1" calls "$(NF-5)" avg_ns "$(NF-4)" min "$(NF-2)" max "$(NF-1)}' | cut -c1-200
  rm -rf $OUT/s
done
