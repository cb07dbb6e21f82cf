#!/bin/bash
# A/B of library variants with tools/ablate.py (frame-loop us per frame, best of 4 batches per process), REPS processes each,
# alternating: tools/ablate_ab.sh "ENV=.. ENV=.." lib1.so lib2.so ...   ("default" = the tree's library)
ENVS=$1; shift
REPS=${REPS:-4}
for rep in $(seq $REPS); do
  for lib in "$@"; do
    l=$lib; [ "$lib" = default ] && l=""
    r=$(env $ENVS ADDER_HIP_LIB=$l python tools/ablate.py 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['us_per_frame'])")
    echo "$lib $r"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END {for (k in a) print k":"a[k]}'
