#!/bin/bash
# the run-records kernel (adder_rr_kernel): the GPU tests that reach it (crf 0, Collapse, delta_t_max > time_spanned) and
# the default-mode legs of the bench with it, with adder_cr_kernel (ADDER_HIP_NO_RR=1) and with adder_cb_kernel
OUT=gpurun_out/rr; mkdir -p $OUT
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_host_mirror.py -x -q -m gpu \
  -k "default_mode or run_records or crf0 or cb_kernel or fuzz or full_size or capacity or eager_and_graph or batch_lengths or model_fixtures or quality_change or long_static or second_opinion" > $OUT/tests_full.txt 2>&1
tail -8 $OUT/tests_full.txt | tee $OUT/tests.txt
legs() {
  env $1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --skip-roofline 2>$OUT/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for l in d['secondary']:
    if 'dtm 7650' in l['workload'] or 'default' in l['workload']:
        print('$1', l['workload'][:70], l.get('us_per_frame'), l.get('value'), l.get('frac'), l.get('error'))
"
}
legs "A=1" | tee $OUT/legs_rr.txt
legs "ADDER_HIP_NO_RR=1" | tee $OUT/legs_cr.txt
