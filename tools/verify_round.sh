#!/bin/bash
# One GPU call's worth of end-of-milestone checks: -m gpu tests, the default bench line, the reference-default
# (generic) mode with the shipped library vs a variant (BASE_LIB), then the round's rocprofv3 set.
OUT=gpurun_out/${1:-verify}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; grep -E "passed|failed|error" $OUT/gputest.log | tail -3
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_one_frame_per_launch']['frac'], d['roofline_one_frame_per_launch']['launch_avg_us'])"
for lib in "${BASE_LIB:-}" ""; do
  for args in "--delta-t-max 7650 --time-mode absolute_t" "--delta-t-max 7650"; do
    r=$(env ${lib:+ADDER_HIP_LIB=$lib} python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-roofline --no-end-to-end $args 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "=== lib=${lib:-shipped} $args: $r"
  done
done
tools/profile_round.sh ${2:-r02} > $OUT/profile_round.log 2>&1
