#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms/step', round(j['ms_per_step'],2), {k:round(v,1) for k,v in j['stages_ms_per_step'].items()})"; }
run A=1
run LDW_NO_STREAM_PRIO=1
run LDW_APX_TILE=22
run LDW_APX_TILE=24
