ulimit -c 0
for b in 0 1; do BATCH=$b timeout 100 python tests/tools_framer.py 2>&1 | tail -1 | cut -c1-200; done
BATCH=1 MULTI=1 TMODE=1 DTM=7650 timeout 100 python tests/tools_framer.py 2>&1 | tail -1 | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_framer.py -x -q 2>&1 | tail -2
