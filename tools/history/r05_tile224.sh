#!/bin/bash
# r05: 2 x 2 wave tiles of gemm_apx_kernel at FOUR waves per SIMD (LDW_APX_TILE=224, experiments build) with and without the SDWA table index
# (gpurun_var_e0.so = experiments build as shipped, gpurun_var_e1.so = + -DLDW_APX_SDWA, default scheduler): full launches without the table epilogue
# (LDW_NO_FUSE_TAB=1, --no-prune), per-launch time from the bench's serialized replay
cd "$GRAFT_REPO_ROOT"
export LDW_NO_FUSE_TAB=1
for rep in 1 2; do
for lib in e0 e1; do
for t in 42 22 224; do
  LDW_AMD_LIB=$PWD/gpurun_var_$lib.so LDW_APX_TILE=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --no-prune 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('lib $lib tile $t', 'ms_per_step', round(d['ms_per_step'], 2), 'gemm avg launch ms', round(r['avg_launch_ms'], 4), 'frac', round(r['frac'], 3), 'launches', r['launches'], d['links'], 'misses', d['spec_misses'])"
done
done
done
