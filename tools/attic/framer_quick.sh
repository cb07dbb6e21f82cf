#!/bin/bash
# kernel time of the framer tiles kernel for a few env settings: tools/attic/framer_quick.sh "ENV=.." "ENV=.." ...
REPO=$(pwd); OUT=$REPO/gpurun_out/fq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for envs in "$@"; do
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o b -- python $REPO/tools/framer_bench.py > $OUT/log.txt 2>&1
  f=$(find $OUT/s -name '*kernel_stats.csv' | head -1)
  echo "=== $envs: $(grep tiles_kernel $f | awk -F, '{print $(NF-5)" calls avg_ns "$(NF-4)}' | head -1)  $(grep slices_kernel $f | awk -F, '{print "slices avg_ns "$(NF-4)}')"
  rm -rf $OUT/s
done
