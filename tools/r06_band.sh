#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bounds.py -x -q -m gpu -k "full_size_properties or c2_full or spans_equal or apx_path or sr_only or fuzz_paths or golden or adversarial or unsorted" > gpurun_out/r06_band_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r06_band_tests.log
bash tools/prof_run.sh r06_band_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
grep -E "gemm_bits|gemm_apx" gpurun_out/r06_band_serial_kernel_stats.csv
for rep in 1 2 3; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'], 2), 'serial stages', {k: round(v, 2) for k, v in d['stages_ms_per_step'].items()}, d['links'])"; done
