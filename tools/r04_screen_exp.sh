#!/bin/bash
# measurement: what the span screen's time is made of (LDW_SCREEN_EXP: 1 no table path, 2 no cell path, 3 neither); serial kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for e in 0 1 2 3; do
  LDW_SCREEN_EXP=$e bash tools/prof_run.sh scrx$e --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "== exp $e"; grep -E "k_mi_screen|gemm_apx" gpurun_out/scrx${e}_kernel_stats.csv
done
