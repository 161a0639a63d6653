#!/bin/bash
# round-2 GPU call: regrowth test, driver-style bench line, kernel stats (overlapped + serial), PMC traffic, hist-vs-GEMM
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "regrow or lr_tukey or end_to_end" > gpurun_out/r02_t2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_t2.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a_c4_bench.json 2> gpurun_out/r02a_c4_bench.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r02a_c4_bench.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r02a_c4_bench.json") if l.startswith("{")][0])
print({k: j[k] for k in ("value","ms_per_step","sustained","plain","cold_first_pass_ms","first_pass_incl_allocations_ms","links","cold_links") if k in j})
print({k: v for k, v in j["roofline"].items() if k not in ("note",)})
print(j.get("cpu_baseline")); print(j["stages_ms_per_step"], j["counters"], j["counters_replay"])
PY
bash tools/prof_run.sh r02a_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh r02a_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"; head -30 gpurun_out/r02a_c4_serial_kernel_stats.csv
bash tools/pmc_traffic2.sh gpurun_out/r02_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/r02_pmc_traffic.log 2>&1; echo "pmc rc $?"; tail -30 gpurun_out/r02_pmc_traffic.log
# hist vs GEMM, plain path (every pair fp64), reduced block list: 20k SNPs x 5k seqs = 2 diagonal + 1 off-diagonal 10k block
bash tools/prof_run.sh r02_hist_vs_gemm_hist --engine hist --L 20000 --N 5000 --steps 1 --warmup 1 --no-mixed --screen 0 --path 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof hist rc $?"
bash tools/prof_run.sh r02_hist_vs_gemm_mfma --engine mfma --L 20000 --N 5000 --steps 1 --warmup 1 --no-mixed --screen 0 --path 1 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof mfma rc $?"
bash tools/pmc_traffic2.sh gpurun_out/r02_hist_vs_gemm_hist_pmc.json --engine hist --L 20000 --N 5000 --steps 1 --warmup 0 --no-mixed --screen 0 --path 1 --no-overlap > gpurun_out/r02_hvg_h.log 2>&1; echo "pmc hist rc $?"; tail -12 gpurun_out/r02_hvg_h.log
bash tools/pmc_traffic2.sh gpurun_out/r02_hist_vs_gemm_mfma_pmc.json --engine mfma --L 20000 --N 5000 --steps 1 --warmup 0 --no-mixed --screen 0 --path 1 --no-overlap > gpurun_out/r02_hvg_m.log 2>&1; echo "pmc mfma rc $?"; tail -12 gpurun_out/r02_hvg_m.log
