#!/bin/bash
# round-3 quick loop on the GPU box: a pytest selection ($K), then the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r03q}
if [ -n "$K" ]; then
  timeout -k 10 ${TEST_TIMEOUT:-700} python -m pytest tests -m gpu -x -q --durations=8 -k "$K" > gpurun_out/${T}_tests.log 2>&1; rc=$?
  echo "pytest rc $rc"; tail -15 gpurun_out/${T}_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 400 python bench.py --steps ${STEPS:-10} --warmup 2 $BENCH_FLAGS > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; rc=$?; echo "bench rc $rc"
[ $rc -ne 0 ] && tail -30 gpurun_out/${T}_bench.err
python - "$T" <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/{sys.argv[1]}_bench.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value","ms_per_step","spec_misses","links")})
for k in ("warm_replay","sustained","mi_values_produced","job","path","prune","cpu_baseline","stages_ms_per_step","stages_ms_per_step_overlapped","counters"):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ("what","note","sample")}
    print(k, v)
print({k: v for k, v in j["roofline"].items() if k not in ("note","measured_in","traffic_source")})
PY
