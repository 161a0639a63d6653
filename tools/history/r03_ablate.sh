#!/bin/bash
# timing ablations of k_mi_screen (separate build directories; results of an ablated build are WRONG, only the kernel time is read):
#   logs   the two v_log_f32 per cell replaced by a subtraction
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in none LDW_ABLATE_SCREEN_LOGS; do
  dir=$PWD/gpurun_out/abl_$v; mkdir -p $dir
  extra=""; [ $v != none ] && extra="-D$v"
  make -C ldweaver_amd/csrc -j16 OBJDIR=$dir OUT=$dir/libldweaver_amd.so "CXXFLAGS=-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $extra" > gpurun_out/abl_$v.log 2>&1 || { echo "build $v failed"; tail -3 gpurun_out/abl_$v.log; continue; }
  LDW_AMD_LIB=$dir/libldweaver_amd.so bash tools/prof_run.sh r03abl_$v --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap --warm > /dev/null 2>&1
  echo "$v: $(grep -E 'k_mi_screen' gpurun_out/r03abl_${v}_kernel_stats.csv | tr '\n' ' ')"
  rm -rf $dir
done
