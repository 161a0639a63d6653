#!/bin/bash
# round-3 evidence at the committed code (parts selected by $PARTS, default all): driver-style bench line, kernel stats (overlapped +
# serial), PMC traffic, SQ counters of the GEMM variants, the histogram-vs-GEMM A/B, flag matrix on the big shapes, whole-job stages
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r03}
PARTS=${PARTS:-bench prof pmc gemm hist flags e2e}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_c4_driver_bench.json 2> gpurun_out/${T}_c4_driver_bench.err; echo "bench rc $?"
python - "$T" <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/{sys.argv[1]}_c4_driver_bench.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value","ms_per_step","spec_misses","links")})
for k in ("warm_replay","sustained","mi_values_produced","job","cpu_baseline","stages_ms_per_step"):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ("what","note","sample")}
    print(k, v)
print({k: v for k, v in j["roofline"].items() if k not in ("note","measured_in","traffic_source")})
PY
fi
if has prof; then
bash tools/prof_run.sh ${T}_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh ${T}_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"; head -30 gpurun_out/${T}_c4_serial_kernel_stats.csv
fi
if has pmc; then
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
fi
if has gemm; then
bash tools/pmc_kernel2.sh "gemm_apx|k_mi_screen|k_pair|k_sel" gpurun_out/${T}_pmc_sq.json --no-extra-legs > gpurun_out/${T}_pmc_sq.log 2>&1; echo "pmc sq rc $?"
LDW_APX_KERNEL=lds bash tools/pmc_kernel2.sh "gemm_apx" gpurun_out/${T}_pmc_sq_ldsgemm.json --no-extra-legs > gpurun_out/${T}_pmc_sq_ldsgemm.log 2>&1; echo "pmc sq (lds gemm) rc $?"
LDW_APX_GRAN=1 bash tools/pmc_kernel2.sh "gemm_apx" gpurun_out/${T}_pmc_sq_finegemm.json --no-extra-legs > gpurun_out/${T}_pmc_sq_finegemm.log 2>&1; echo "pmc sq (fine exponents) rc $?"
fi
if has hist; then
# the north-star A/B: every pair's fp64 MI on the plain path (no screen, no approximate GEMM), 2 diagonal + 1 off-diagonal 10k block
HB="--L 20000 --N 5000 --steps 2 --warmup 1 --no-mixed --screen 0 --path 1 --no-overlap --no-cpu-baseline --no-extra-legs --warm"
for e in hist mfma hist_states; do
  st="--steps 2"; [ $e = hist_states ] && st="--steps 1 --warmup 0"
  bash tools/prof_run.sh ${T}_hist_vs_gemm_${e} $HB $st --engine $e > /dev/null 2>&1; echo "hist A/B prof $e rc $?"; head -5 gpurun_out/${T}_hist_vs_gemm_${e}_kernel_stats.csv
  [ $e = hist_states ] && continue
  bash tools/pmc_traffic2.sh gpurun_out/${T}_hist_vs_gemm_${e}_pmc.json $HB --engine $e > gpurun_out/${T}_hist_vs_gemm_${e}_pmc.log 2>&1; echo "hist A/B pmc $e rc $?"
done
fi
if has flags; then
timeout -k 10 400 python tools/flag_matrix.py --base "--L 85000 --N 616 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs" --only 0,1,2,6,8,9,13 --out gpurun_out/${T}_flags_c3.json > gpurun_out/${T}_flags_c3.log 2>&1; echo "matrix c3 rc $?"; tail -1 gpurun_out/${T}_flags_c3.log
timeout -k 10 600 python tools/flag_matrix.py --base "--L 500000 --N 10000 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs" --only 0,8,9 --out gpurun_out/${T}_flags_c5.json > gpurun_out/${T}_flags_c5.log 2>&1; echo "matrix c5 rc $?"; tail -1 gpurun_out/${T}_flags_c5.log
fi
if has e2e; then
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 500 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"; cat gpurun_out/${T}_e2e_c5_1gpu_stages.json
fi
