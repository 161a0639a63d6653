#!/bin/bash
# one approximate-GEMM kernel variant ($1 = lds|reg): parity tests of the approximate path, serial kernel profile
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
k=${1:-lds}; tag=${2:-r03one}
LDW_APX_KERNEL=$k timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "apx or table_test" > gpurun_out/${tag}_${k}_tests.log 2>&1; rc=$?
echo "$k pytest rc $rc"; tail -3 gpurun_out/${tag}_${k}_tests.log
[ $rc -ne 0 ] && exit $rc
LDW_APX_KERNEL=$k bash tools/prof_run.sh ${tag}_${k}_serial --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "$k prof rc $?"
head -6 gpurun_out/${tag}_${k}_serial_kernel_stats.csv
