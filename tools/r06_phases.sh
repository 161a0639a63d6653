#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ph in 1 2 3; do
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-adversarial --no-job --sustain-s 0 --gather-phases $ph 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
sm = d['scaling_model']['predicted']
print('gather-phases $ph:', {n: (v['phases'], round(v['slowest_share_compute_ms'], 2), round(v['exposed_gather_model_ms'], 2), round(v['predicted_ms_per_step'], 2)) for n, v in sm.items()})"
done
