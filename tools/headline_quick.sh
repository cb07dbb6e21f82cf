#!/bin/bash
# the headline path after a kernel change: the GPU tests that reach the lean kernels, then the headline line twice
OUT=gpurun_out/hq; mkdir -p $OUT
if [ "${1:-}" != "notest" ]; then
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_records.py -x -q -m gpu \
  -k "crf0 or lean or fuzz or full_size or capacity or eager_and_graph or batch_lengths or model_fixtures or records or band or misalign or ragged" > $OUT/tests_full.txt 2>&1
tail -3 $OUT/tests_full.txt
fi
for i in 1 2; do
  python bench.py --steps 32 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('headline', d['ms_per_step'], d['value'], r['frac'], r['frame_kernel_launch_us'], r['scan_offsets_expand_us'], r['chunk_us'])"
done | tee $OUT/headline.txt
