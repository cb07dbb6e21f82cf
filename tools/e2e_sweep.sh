#!/bin/bash
# the per-frame ring: workgroups of the wire hand-over (ADDER_HIP_WIRE_BLOCKS) and of the AdderEvent copy
# (ADDER_HIP_OUT_BLOCKS); then kernel stats of the default-quality leg (what a frame costs on the device at T = 1)
OUT=gpurun_out/e2e; mkdir -p $OUT
for wb in 64 128 256 512 1024; do
  echo "== ADDER_HIP_WIRE_BLOCKS=$wb ADDER_HIP_OUT_BLOCKS=$wb"
  ADDER_HIP_WIRE_BLOCKS=$wb ADDER_HIP_OUT_BLOCKS=$wb python tools/e2e_probe.py ring 2>/dev/null | grep per_frame
done | tee $OUT/ring_sweep.txt
python tools/e2e_probe.py dq 2>/dev/null | grep default_quality | tee $OUT/dq.txt
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/ks -o dq -- python $REPO/tools/e2e_probe.py dq > $REPO/$OUT/dq_prof.log 2>&1
f=$(find $REPO/$OUT/ks -name '*kernel_stats.csv' | head -1); cp $f $REPO/$OUT/dq_kernel_stats.csv; rm -rf $REPO/$OUT/ks
cut -c1-150 $REPO/$OUT/dq_kernel_stats.csv | head -14
