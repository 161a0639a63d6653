#!/bin/bash
# r05: the 3 x 2 wave tile of gemm_apx_kernel (three waves per SIMD) against 4 x 2 / 2 x 2, experiments build, full launches without the table
# epilogue (LDW_NO_FUSE_TAB=1, --no-prune): per-launch time of the kernel from the bench's serialized replay + link counts
cd "$GRAFT_REPO_ROOT"
export LDW_AMD_LIB=$PWD/ldweaver_amd/libldweaver_amd_exp.so LDW_NO_FUSE_TAB=1
for rep in 1 2; do
for t in 42 32 22; do
  LDW_APX_TILE=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --no-prune 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('tile $t', 'ms_per_step', round(d['ms_per_step'], 2), 'gemm avg launch ms', round(r['avg_launch_ms'], 4), 'frac', round(r['frac'], 3), 'launches', r['launches'], d['links'], 'misses', d['spec_misses'])"
done
done
