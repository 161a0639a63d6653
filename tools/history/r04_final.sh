#!/bin/bash
# round-4 evidence at the committed code (parts selected by $PARTS, default all): driver-style bench line, kernel stats (overlapped +
# serial), overlapped timeline, PMC traffic, SQ counters of the GEMM and the screens, whole-job stages at C4 and C5 (+ kernel stats of
# the C5 job), first-pass probe, cold job profile, flag matrix on C5
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r04}
PARTS=${PARTS:-bench prof timeline pmc sq e2e first flags}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_c4_driver_bench.json 2> gpurun_out/${T}_c4_driver_bench.err; echo "bench rc $?"
python - "$T" <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/{sys.argv[1]}_c4_driver_bench.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value","ms_per_step","spec_misses","links","first_pass_incl_allocations_ms")})
print("job", {k: v for k, v in (j.get("job") or {}).items() if k in ("job_s","stages_s","h2d_ms")})
print({k: v for k, v in j["roofline"].items() if k in ("achieved","frac","avg_launch_ms","traffic")})
PY
fi
if has prof; then
bash tools/prof_run.sh ${T}_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh ${T}_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"; head -30 gpurun_out/${T}_c4_serial_kernel_stats.csv
fi
if has timeline; then
bash tools/r04_timeline.sh ${T} > /dev/null 2>&1; echo "timeline rc $?"; head -4 gpurun_out/${T}_timeline.txt
fi
if has pmc; then
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
fi
if has sq; then
bash tools/pmc_kernel2.sh "gemm_apx|k_mi_screen|k_pair|k_sel|gemm_bits" gpurun_out/${T}_pmc_sq.json --no-extra-legs > gpurun_out/${T}_pmc_sq.log 2>&1; echo "pmc sq rc $?"
fi
if has e2e; then
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 500 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"; cat gpurun_out/${T}_e2e_c5_1gpu_stages.json
bash tools/prof_cmd.sh ${T}_e2e_c5 tools/e2e_bench.py --L 500000 --N 10000 --repeat 1 > /dev/null 2>&1; echo "e2e c5 prof rc $?"
fi
if has first; then
LDW_HOST_TIMING=1 timeout -k 10 300 python tools/first_pass_probe.py > gpurun_out/${T}_first_pass_probe.txt 2>&1; echo "first rc $?"; grep -E "cold pass|set_alignment" gpurun_out/${T}_first_pass_probe.txt
LDW_HOST_TIMING=1 timeout -k 10 300 python tools/job_profile.py --cold > gpurun_out/${T}_job_profile_cold.txt 2>&1; echo "job profile rc $?"
fi
if has flags; then
timeout -k 10 600 python tools/flag_matrix.py --base "--L 500000 --N 10000 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs" --only 0,8,9 --out gpurun_out/${T}_flags_c5.json > gpurun_out/${T}_flags_c5.log 2>&1; echo "matrix c5 rc $?"; tail -1 gpurun_out/${T}_flags_c5.log
fi
