#!/bin/bash
# Builds ONE variant of ldw_apx.hip into its own library gpurun_var_<tag>.so at the repo root (the other objects are taken from the last `make`
# of the same flavour), for the A/B scripts tools/r05_sdwa_ab.sh / tools/r05_tile224.sh:
#     tools/r05_build_variant.sh <tag> "<defines>" [exp] [scheduler]
#     e.g.  v0 ""   |   v2 "-DLDW_APX_SDWA"   |   v3 "-DLDW_APX_SDWA -DLDW_APX_PREFETCH"   |   lut16 "-DLDW_APX_LUT16"   |   mt2 "-DLDW_APX_MT=2"
#           e0 "" exp   |   e1 "-DLDW_APX_SDWA" exp default     (iterative-ilp crashes hipcc 7.2 on the experiments build with -DLDW_APX_SDWA)
# The libraries are scratch: delete them after the run (they travel to the GPU box with the snapshot).
set -e
tag=$1; defs=$2; flavour=${3:-lean}; sched=${4:-iterative-ilp}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/ldweaver_amd/csrc"
mkdir -p "$root/build/var"
objdir=obj; extra=""
if [ "$flavour" = exp ]; then objdir=obj_exp; extra="-DLDW_EXPERIMENTS"; fi
schedflag="-mllvm -amdgpu-sched-strategy=$sched"
[ "$sched" = default ] && schedflag=""
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --offload-compress -Wall -Wno-unused-function -Wno-unused-result $extra $schedflag $defs -c ldw_apx.hip -o "$root/build/var/ldw_apx_$tag.o"
objs=$(ls "$root/build/$objdir"/*.o | grep -v ldw_apx.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/gpurun_var_$tag.so" $objs "$root/build/var/ldw_apx_$tag.o"
ls -la "$root/gpurun_var_$tag.so"
