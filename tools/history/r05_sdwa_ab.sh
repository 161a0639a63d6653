#!/bin/bash
# r05: variants of gemm_apx_kernel's table-index arithmetic (SDWA byte select) and read placement (prefetch), one library per variant
# (gpurun_var_<tag>.so built by hand from ldw_apx.hip with -DLDW_APX_NO_SDWA / -DLDW_APX_PREFETCH): per-launch time of the kernel from the
# bench's serialized replay (roofline.avg_launch_ms), step time, link counts.  Two rounds, alternating.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for t in "$@"; do
  LDW_AMD_LIB=$PWD/gpurun_var_$t.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('variant $t', 'ms_per_step', round(d['ms_per_step'], 2), 'gemm avg launch ms', round(r['avg_launch_ms'], 4), 'overlapped', round(r['overlapped_avg_launch_ms'], 4), 'frac', round(r['frac'], 3), 'launches', r['launches'], d['links'], 'misses', d['spec_misses'])"
done
done
