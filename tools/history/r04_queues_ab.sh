#!/bin/bash
# A/B of the queue assignment (r04 end): default = screens at the head of phase 2 on the main stream + exact band GEMM on the GEMM stream;
# LDW_SCREEN_GEMMQ=1 LDW_BAND_LATE=1 = the earlier places.  Alternating runs, C4 and the 85k x 616 shape; C5 once each (misses!).
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; env "$@" python bench.py $ARGS --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', round(j['ms_per_step'],2), 'misses', j['spec_misses'], j['links'], 'pairs', j['counters']['apx_pairs_listed'])"; }
ARGS="--steps 20 --warmup 3"
for i in 1 2 3; do run "C4 new" X=1; run "C4 old" LDW_SCREEN_GEMMQ=1 LDW_BAND_LATE=1; done
ARGS="--L 85000 --N 616 --steps 20 --warmup 3"
for i in 1 2; do run "616 new" X=1; run "616 old" LDW_SCREEN_GEMMQ=1 LDW_BAND_LATE=1; done
ARGS="--L 500000 --N 10000 --steps 2 --warmup 1"
run "C5 new" X=1; run "C5 old" LDW_SCREEN_GEMMQ=1 LDW_BAND_LATE=1
