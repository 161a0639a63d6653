#!/bin/bash
# r05 A/B: the table-based fp64 logarithm (-DLDW_LOG_TABLE build in ldweaver_amd/libldweaver_amd_lt.so) against the reciprocal + atanh series:
# plain path (fp64 MI of every pair) and default path, serial kernel times
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in libldweaver_amd.so libldweaver_amd_lt.so; do
  for mode in "--no-mixed --screen 0 --path 1" ""; do
    LDW_AMD_LIB=$PWD/ldweaver_amd/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs $mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '[$mode]', 'ms_per_step', round(d['ms_per_step'], 2), 'serial stages', {k: round(v, 2) for k, v in d['stages_ms_per_step'].items()}, d['links'])"
  done
done
done
