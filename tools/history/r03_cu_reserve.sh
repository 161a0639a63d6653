#!/bin/bash
# experiment: CUs reserved for the main stream's small kernels (LDW_CU_RESERVE = modulus of the CUs the GEMM stream may not use)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for m in 0 17 9 5; do
  LDW_CU_RESERVE=$m timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/r03cu_$m.json 2> gpurun_out/r03cu_$m.err || { echo "m=$m failed"; tail -3 gpurun_out/r03cu_$m.err; exit 1; }
  python - $m <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/r03cu_{sys.argv[1]}.json") if l.startswith("{")][0])
print("reserve modulus", sys.argv[1], "cold ms", round(j["ms_per_step"],2), "gemm avg (serial replay)", round(j["roofline"]["avg_launch_ms"],4), "overlapped", round(j["roofline"]["overlapped_avg_launch_ms"],4), j["links"])
PY
done
