#!/bin/bash
for v in "" "--no-mixed"; do
  echo "== $v"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 $v 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step', round(d['ms_per_step'],1), 'stages', {k: round(v,1) for k,v in d['stages_ms_per_step'].items()}, d['counters'], d['links'])
"
done
