#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
dir=$PWD/gpurun_out/var_stats; mkdir -p $dir
make -C ldweaver_amd/csrc -j16 OBJDIR=$dir OUT=$dir/libldweaver_amd.so "CXXFLAGS=-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DLDW_SCREEN_STATS" > gpurun_out/var_stats.log 2>&1 || { echo "build failed"; grep -E "error" gpurun_out/var_stats.log | head; exit 1; }
for k in survey adversarial; do LDW_AMD_LIB=$dir/libldweaver_amd.so python tools/screen_stats_probe.py $k 2>&1 | grep -E "pairs|per pair"; done
rm -rf $dir
