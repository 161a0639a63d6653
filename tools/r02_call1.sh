#!/bin/bash
# round-2 GPU call 1: baseline tests, flag matrix (reproduce the r01 reserve_keep failure), hist-vs-GEMM profiles
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout -k 10 420 python -m pytest tests -m gpu -x -q > gpurun_out/r02_s1_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02_s1_gputests.log )
echo "tests done"; tail -3 gpurun_out/r02_s1_gputests.log
timeout -k 10 400 python tools/flag_matrix.py --only 0,1,2,3,4,5,6,7 --out gpurun_out/r02_flags_small.json > gpurun_out/r02_flags_small.log 2>&1; echo "matrix small rc $?"
timeout -k 10 300 python tools/flag_matrix.py --base "--L 85000 --N 616 --steps 2 --warmup 1 --no-cpu-baseline" --only 0,1,6 --out gpurun_out/r02_flags_c3.json > gpurun_out/r02_flags_c3.log 2>&1; echo "matrix c3 rc $?"
# hist vs GEMM on the plain path (every pair fp64), reduced block list: 20k SNPs = 2 diagonal + 1 off-diagonal block of 10k
bash tools/prof_run.sh r02_hist_vs_gemm_hist --engine hist --L 20000 --N 5000 --steps 1 --warmup 1 --no-mixed --screen 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1; echo "prof hist rc $?"
bash tools/prof_run.sh r02_hist_vs_gemm_mfma --engine mfma --L 20000 --N 5000 --steps 1 --warmup 1 --no-mixed --screen 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1; echo "prof mfma rc $?"
timeout -k 10 500 python bench.py --L 500000 --N 10000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02_c5_bench.log 2>&1; echo "c5 bench rc $?"
tail -c 600 gpurun_out/r02_c5_bench.log
