#!/bin/bash
# A/B of environment switches on the default bench: usage tools/r03_ab.sh "VAR1=x" "VAR2=y" ...  ("" = defaults)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-job > gpurun_out/r03ab_$i.json 2> gpurun_out/r03ab_$i.err || { echo "[$e] failed"; tail -3 gpurun_out/r03ab_$i.err; exit 1; }
  python - "$i" "$e" <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/r03ab_{sys.argv[1]}.json") if l.startswith("{")][0])
print(f"[{sys.argv[2]}] cold {j['ms_per_step']:.2f} ms  warm {j.get('warm_replay',{}).get('ms_per_step',0):.2f}  sustained {j.get('sustained',{}).get('ms_per_step',0):.2f}  misses {j['spec_misses']}  links {j['links']}")
PY
done
