#!/bin/bash
# timing experiment: screen kernel with one class of units skipped (results are wrong on purpose; only kernel times are read)
cd "$GRAFT_REPO_ROOT"
for e in 1 2; do
  touch ldweaver_amd/csrc/ldw_mi.hip
  make -C ldweaver_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DLDW_EXP=$e" > gpurun_out/exp_build_$e.log 2>&1 || { tail -5 gpurun_out/exp_build_$e.log; exit 1; }
  bash tools/prof_run.sh r02exp$e --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "exp $e rc $?"
  grep -E "k_mi_screen|gemm_apx|k_pair" gpurun_out/r02exp${e}_kernel_stats.csv
done
