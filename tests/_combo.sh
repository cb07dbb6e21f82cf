run() { echo "== $*"; env "$@" timeout 200 python tests/tools_ablate.py 2>&1 | grep -v "Extension modules" | tail -1; }
run T=256
run T=300
run T=256 ADDER_HIP_NO_GRAPH=1
run T=256 CONTENT=1
run T=64 W=3840 H=2160
run T=64 MULTI=0
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline | tail -1 | cut -c1-400
