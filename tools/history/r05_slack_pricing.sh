#!/bin/bash
# What an ABSOLUTE slack per entry of the approximate GEMM would cost the screen (experiments build, LDW_APX_EXTRA_UNITS = weight units): listed pairs, pruned tiles and the
# pass time at C4 — the price list for a contraction over compressed clone groups.  Results stay exact for every value (the bounds only loosen).
cd "$GRAFT_REPO_ROOT"
export LDW_AMD_LIB=$PWD/ldweaver_amd/libldweaver_amd_exp.so
for x in 0 0.005 0.01 0.02 0.03 0.06 0.12; do
  LDW_APX_EXTRA_UNITS=$x timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters_replay']; pr=d['prune']; st=d['stages_ms_per_step']
print('extra_units $x  ms_per_step %.2f  pairs_listed/pass %d  units_listed/pass %d  tiles_pruned %.3f  serial gemm %.2f epilogue %.2f  links %s' % (d['ms_per_step'], c['apx_pairs_listed']/3, c['apx_units_listed']/3, pr['tiles_pruned']/max(1,pr['tiles_total']), st['gemm_ms'], st['epilogue_ms'], d['links']))"
done
