#!/bin/bash
# A/B of the maybe list (LDW_NO_MAYBE=1: table-eligible regions with a failing entry are stored and screened whole, as in r03)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for e in off on; do
  [ $e = off ] && export LDW_NO_MAYBE=1 || unset LDW_NO_MAYBE
  bash tools/prof_run.sh maybe_$e --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "== maybe $e"; grep -E "k_mi_screen|gemm_apx|k_screen_maybe|k_pair" gpurun_out/maybe_${e}_kernel_stats.csv
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step', round(j['ms_per_step'],2), 'misses', j['spec_misses'], j['links'], 'pairs', j['counters']['apx_pairs_listed'], 'frac', round(j['roofline']['frac'],4))"
done
