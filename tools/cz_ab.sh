#!/bin/bash
# A/B of the lazy-levels kernel (adder_cz_kernel) against the bounded Collapse kernel (ADDER_HIP_NO_CZ=1) with tools/ablate.py:
# frame-loop us per frame, best of ITERS batches, over the contents the regime sees.  tools/cz_ab.sh [lib.so]
LIB=${1:-}
export ITERS=${ITERS:-6} MULTI=1 TMODE=1 DTM=7650
run() {  # name, env...
  name=$1; shift
  for off in 0 1; do
    r=$(env "$@" ADDER_HIP_NO_CZ=$off ADDER_HIP_LIB=$LIB python tools/ablate.py 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['us_per_frame'], d['events_per_unit_frame'])")
    echo "$name no_cz=$off us_per_frame,events_per_unit_frame: $r"
  done
}
run scene_crf3   CONTENT=2 CRF=2,7,7 T=300
run static_crf3  CONTENT=0 CRF=2,7,7 T=300
run noise_crf3   CONTENT=1 CRF=2,7,7 T=120
run scene_crf9   CONTENT=2 CRF=15,25,1 T=300
run scene_crf3_dt CONTENT=2 CRF=2,7,7 T=300 TMODE=0
run c5_4k_rgb    CONTENT=2 CRF=2,7,7 T=64 W=3840 H=2160 C=3
