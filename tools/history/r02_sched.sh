#!/bin/bash
# One translation unit under different LLVM AMDGPU scheduler strategies (kernel times from a serial profile of the bench).
# Every strategy is built into its OWN object directory and library (gpurun_out/sched_<strategy>/) and run through LDW_AMD_LIB,
# so the default build (build/obj + ldweaver_amd/libldweaver_amd.so) is never touched.
#   usage (GPU box): FILE=ldw_apx PAT=gemm_apx tools/r02_sched.sh max-ilp iterative-ilp ...
cd "$GRAFT_REPO_ROOT"
FILE=${FILE:-ldw_apx}; PAT=${PAT:-gemm_apx}
for strat in "$@"; do
  extra="-mllvm -amdgpu-sched-strategy=$strat"; [ "$strat" = default ] && extra=""
  dir=$PWD/gpurun_out/sched_$strat; mkdir -p $dir
  make -C ldweaver_amd/csrc -j8 OBJDIR=$dir OUT=$dir/libldweaver_amd.so > gpurun_out/sched_$strat.log 2>&1 || { echo "$strat: base build failed"; continue; }
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $extra -c ldweaver_amd/csrc/$FILE.hip -o $dir/$FILE.o >> gpurun_out/sched_$strat.log 2>&1 || { echo "$strat: compile failed"; tail -2 gpurun_out/sched_$strat.log; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $dir/libldweaver_amd.so $dir/*.o || continue
  LDW_AMD_LIB=$dir/libldweaver_amd.so bash tools/prof_run.sh r02s_$strat --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1
  echo "$strat: $(grep -E "$PAT" gpurun_out/r02s_${strat}_kernel_stats.csv | tr '\n' ' ')"
  rm -rf $dir
done
