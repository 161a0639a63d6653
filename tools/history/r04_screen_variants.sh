#!/bin/bash
# k_mi_screen compile-time variants built on the box in separate directories (the default build is untouched).
# usage: tools/r04_screen_variants.sh name=-DFLAG[,-DFLAG2] ...      ("base=" = the in-tree library)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%=*}; flags=$(echo "${spec#*=}" | tr ',' ' ')
  if [ -n "$flags" ]; then
    dir=$PWD/gpurun_out/var_$name; mkdir -p $dir
    make -C ldweaver_amd/csrc -j16 OBJDIR=$dir OUT=$dir/libldweaver_amd.so "CXXFLAGS=-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $flags" > gpurun_out/var_$name.log 2>&1 || { echo "build $name failed"; tail -3 gpurun_out/var_$name.log; continue; }
    export LDW_AMD_LIB=$dir/libldweaver_amd.so
  else
    unset LDW_AMD_LIB
  fi
  bash tools/prof_run.sh var_$name --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "== $name ($flags)"; grep -E "k_mi_screen<|gemm_apx" gpurun_out/var_${name}_kernel_stats.csv
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step', round(j['ms_per_step'],2), 'misses', j['spec_misses'], j['links'], 'pairs', j['counters']['apx_pairs_listed'])"
  [ -n "$flags" ] && rm -rf $dir
done
