#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in on off; do
  if [ $v = off ]; then export LDW_NO_SCREEN_L1=1; else unset LDW_NO_SCREEN_L1; fi
  bash tools/prof_run.sh "l1_$v" --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "== first level $v"; grep -E "k_mi_screen|k_screen_maybe" gpurun_out/l1_${v}_kernel_stats.csv
done
