#!/bin/bash
# round-2 final evidence at the committed code: driver-style bench line, kernel stats (overlapped + serial), PMC traffic,
# SQ counters of the dominant kernels, flag matrix on the big shapes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r02b}
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_c4_driver_bench.json 2> gpurun_out/${T}_c4_driver_bench.err; echo "bench rc $?"
python - "$T" <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/{sys.argv[1]}_c4_driver_bench.json") if l.startswith("{")][0])
print({k: j[k] for k in ("value","ms_per_step","sustained","cold_first_pass_ms","first_pass_incl_allocations_ms","links") if k in j})
print("plain", {k: v for k, v in j["plain"].items() if k != "what"})
print({k: v for k, v in j["roofline"].items() if k not in ("note",)})
print(j.get("cpu_baseline")); print(j["stages_ms_per_step"], j["counters"])
PY
bash tools/prof_run.sh ${T}_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh ${T}_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"; head -24 gpurun_out/${T}_c4_serial_kernel_stats.csv
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
bash tools/pmc_kernel2.sh "gemm_apx|k_mi_screen|k_pair|k_sel" gpurun_out/${T}_pmc_sq.json --no-extra-legs > gpurun_out/${T}_pmc_sq.log 2>&1; echo "pmc sq rc $?"
timeout -k 10 300 python tools/flag_matrix.py --base "--L 85000 --N 616 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs" --only 0,1,2,6,8,9 --out gpurun_out/${T}_flags_c3.json > gpurun_out/${T}_flags_c3.log 2>&1; echo "matrix c3 rc $?"; tail -1 gpurun_out/${T}_flags_c3.log
timeout -k 10 500 python tools/flag_matrix.py --base "--L 500000 --N 10000 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs" --only 0,8,9 --out gpurun_out/${T}_flags_c5.json > gpurun_out/${T}_flags_c5.log 2>&1; echo "matrix c5 rc $?"; tail -1 gpurun_out/${T}_flags_c5.log
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 400 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"; cat gpurun_out/${T}_e2e_c5_1gpu_stages.json
