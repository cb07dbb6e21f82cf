ulimit -c 0
run() { echo "== $*"; env "$@" timeout 100 python tests/tools_ablate.py 2>&1 | grep -v "Extension modules" | tail -1 | cut -c1-200; }
run T=304 ADDER_HIP_FRAMES_PER_LAUNCH=1
run T=304 ADDER_HIP_FRAMES_PER_LAUNCH=1
run T=304 ADDER_HIP_FRAMES_PER_LAUNCH=1 TMODE=1
run T=304
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_avg_us'], d['roofline_one_frame_per_launch']['frac'], d['roofline_one_frame_per_launch']['launch_avg_us'])"
