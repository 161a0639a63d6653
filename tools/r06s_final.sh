#!/bin/bash
# round 6, second session: final evidence at the committed code — the driver's bench line, kernel stats of the default path (serial + overlapped) and of the plain path,
# PMC traffic + SQ counters of the GEMMs (the scaled panel changed gemm_apx_kernel's instruction mix), the C3 shape, whole-job stages.  Tag r06u (r06t: before the deferred pair-list appends).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=r06u
PLAIN="--no-mixed --screen 0 --path 1"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_c4_driver_bench.json 2> gpurun_out/${T}_c4_driver_bench.err; echo "bench rc $?"
bash tools/prof_run.sh ${T}_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh ${T}_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"
bash tools/prof_run.sh ${T}_c4_plain_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap $PLAIN > /dev/null 2>&1; echo "prof plain serial rc $?"
bash tools/prof_run.sh ${T}_c3shape_serial --L 85000 --N 616 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof c3 serial rc $?"
timeout -k 10 300 python bench.py --L 85000 --N 616 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/${T}_c3shape_bench.json 2>/dev/null; echo "c3 bench rc $?"
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
bash tools/pmc_run.sh "gemm_apx|gemm_bits" gpurun_out/${T}_pmc_gemm.json "" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" > gpurun_out/${T}_pmc_gemm.log 2>&1; echo "pmc sq rc $?"
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 500 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"
