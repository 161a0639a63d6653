#!/bin/bash
# A/B of the two approximate-GEMM kernels (LDW_APX_KERNEL=reg|lds): parity tests of the approximate path, then the serial
# kernel profile of the bench for each
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for k in lds reg; do
  LDW_APX_KERNEL=$k timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "apx or multiallelic or table_test or popcount or mi_blocks_match or twins or edge" > gpurun_out/r03ab_${k}_tests.log 2>&1; rc=$?
  echo "$k pytest rc $rc"; tail -4 gpurun_out/r03ab_${k}_tests.log
  [ $rc -ne 0 ] && exit $rc
  LDW_APX_KERNEL=$k bash tools/prof_run.sh r03ab_${k}_serial --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "$k prof rc $?"
  head -8 gpurun_out/r03ab_${k}_serial_kernel_stats.csv
  LDW_APX_KERNEL=$k timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/r03ab_${k}_bench.json 2> gpurun_out/r03ab_${k}_bench.err; echo "$k bench rc $?"
  python - $k <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/r03ab_{sys.argv[1]}_bench.json") if l.startswith("{")][0])
print(sys.argv[1], "ms/step", round(j["ms_per_step"],2), j["links"], {k:v for k,v in j["roofline"].items() if k in ("kernel","frac","avg_launch_ms","achieved")})
PY
done
