#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for d in 0 1; do
  export LDW_POP_DEBUG=$d
  bash tools/prof_run.sh r02dbg$d --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  echo "== debug $d"; grep "k_units_pop\|k_mi_units_tl" gpurun_out/r02dbg${d}_kernel_stats.csv | head -4
done
