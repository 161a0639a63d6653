#!/bin/bash
# k_mi_screen with V columns in flight per wave in its multi-cell paths (separate build directory; the default build is untouched)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "$@"; do
  dir=$PWD/gpurun_out/scrv_$v; mkdir -p $dir
  make -C ldweaver_amd/csrc -j16 OBJDIR=$dir OUT=$dir/libldweaver_amd.so "CXXFLAGS=-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DLDW_SCREEN_V=$v" > gpurun_out/scrv_$v.log 2>&1 || { echo "build $v failed"; tail -3 gpurun_out/scrv_$v.log; continue; }
  LDW_AMD_LIB=$dir/libldweaver_amd.so bash tools/prof_run.sh r03scrv_$v --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "V=$v: $(grep -E 'k_mi_screen<' gpurun_out/r03scrv_${v}_kernel_stats.csv | tr '\n' ' ')  $(python3 -c "import json;j=json.loads(open('gpurun_out/r03scrv_${v}_bench.json').read());print('ms/step',round(j['ms_per_step'],2),j['links'])")"
  rm -rf $dir
done
