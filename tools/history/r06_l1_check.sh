#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/r06_l1_prof.sh
for rep in 1 2; do for v in "" 1; do
  if [ -n "$v" ]; then export LDW_NO_SCREEN_L1=1; else unset LDW_NO_SCREEN_L1; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NO_L1=[$v]', 'ms_per_step', round(d['ms_per_step'], 2), 'serial stages', {k: round(v, 2) for k, v in d['stages_ms_per_step'].items()}, d['links'], d['path']['pairs_listed'], d['spec_misses'])"
done; done
unset LDW_NO_SCREEN_L1
timeout -k 10 700 python -m pytest tests/test_bounds.py tests/test_gpu_parity.py -x -q -m gpu -k "fuzz_paths or adversarial or table_test or apx_path or c4_blocks or spans_equal or full_size_properties" > gpurun_out/r06_l1_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r06_l1_tests.log
