#!/bin/bash
# round-6 evidence at the committed code (parts selected by $PARTS): the driver's bench command, kernel stats of the default path (overlapped + serial) and
# of the plain path (`mi_values_produced`), PMC traffic of both, SQ counters of the GEMMs and of k_mi_epilogue, the C3 shape, whole-job stages at C4 and C5,
# the in-process multi-context bench.  Everything lands in gpurun_out/ under the tag (copy what is quoted into profiles/).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r06}
PARTS=${PARTS:-bench prof plain c3 pmc sq e2e inproc}
has() { [[ " $PARTS " == *" $1 "* ]]; }
PLAIN="--no-mixed --screen 0 --path 1"
if has bench; then
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_c4_driver_bench.json 2> gpurun_out/${T}_c4_driver_bench.err; echo "bench rc $?"
fi
if has prof; then
bash tools/prof_run.sh ${T}_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
bash tools/prof_run.sh ${T}_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"
fi
if has plain; then
bash tools/prof_run.sh ${T}_c4_plain_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap $PLAIN > /dev/null 2>&1; echo "prof plain serial rc $?"
fi
if has c3; then
bash tools/prof_run.sh ${T}_c3shape_serial --L 85000 --N 616 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof c3 serial rc $?"
timeout -k 10 300 python bench.py --L 85000 --N 616 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/${T}_c3shape_bench.json 2>/dev/null; echo "c3 bench rc $?"
fi
if has pmc; then
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic_plain.json --steps 1 --warmup 1 --no-overlap $PLAIN > gpurun_out/${T}_pmc_traffic_plain.log 2>&1; echo "pmc plain rc $?"
fi
if has sq; then
bash tools/pmc_run.sh "gemm_apx|gemm_bits" gpurun_out/${T}_pmc_gemm.json "" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" > gpurun_out/${T}_pmc_gemm.log 2>&1; echo "pmc sq rc $?"
fi
if has e2e; then
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 500 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"
fi
if has inproc; then
timeout -k 10 300 python bench.py --gpus 2 --inproc --inproc-devices 0,0 --steps 5 --warmup 2 > gpurun_out/${T}_inproc_2ctx_one_gpu.json 2>/dev/null; echo "inproc 2 rc $?"
fi
