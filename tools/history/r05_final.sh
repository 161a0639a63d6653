#!/bin/bash
# round-5 evidence at the committed code (parts selected by $PARTS, default all but cpufull): driver-style bench line, kernel stats of the
# default path (overlapped + serial) and of the plain path (`mi_values_produced`: overlapped + serial), PMC traffic of both, SQ counters of the
# approximate GEMM, whole-job stages at C4 and C5, in-process multi-context bench, the lr-stream probe; cpufull = bench with the stated CPU sample
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T=${TAG:-r05}
PARTS=${PARTS:-bench prof plain pmc sq e2e inproc stream}
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
bash tools/prof_run.sh ${T}_c4_plain --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs $PLAIN > /dev/null 2>&1; echo "prof plain rc $?"
bash tools/prof_run.sh ${T}_c4_plain_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap $PLAIN > /dev/null 2>&1; echo "prof plain serial rc $?"
fi
if has pmc; then
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic.json --steps 1 --warmup 1 --no-overlap > gpurun_out/${T}_pmc_traffic.log 2>&1; echo "pmc rc $?"
bash tools/pmc_traffic2.sh gpurun_out/${T}_pmc_traffic_plain.json --steps 1 --warmup 1 --no-overlap $PLAIN > gpurun_out/${T}_pmc_traffic_plain.log 2>&1; echo "pmc plain rc $?"
fi
if has sq; then
bash tools/pmc_kernel2.sh "gemm_apx|gemm_bits" gpurun_out/${T}_pmc_gemm.json --no-extra-legs > gpurun_out/${T}_pmc_gemm.log 2>&1; echo "pmc sq rc $?"
fi
if has e2e; then
timeout -k 10 300 python tools/e2e_bench.py --L 100000 --N 5000 --out gpurun_out/${T}_e2e_c4_stages.json > /dev/null 2>&1; echo "e2e c4 rc $?"
timeout -k 10 500 python tools/e2e_bench.py --L 500000 --N 10000 --out gpurun_out/${T}_e2e_c5_1gpu_stages.json > /dev/null 2>&1; echo "e2e c5 rc $?"
fi
if has inproc; then
timeout -k 10 300 python bench.py --gpus 1 --inproc --steps 10 --warmup 3 > gpurun_out/${T}_inproc_1ctx.json 2>/dev/null; echo "inproc 1 rc $?"
timeout -k 10 300 python bench.py --gpus 2 --inproc --inproc-devices 0,0 --steps 5 --warmup 2 > gpurun_out/${T}_inproc_2ctx_one_gpu.json 2>/dev/null; echo "inproc 2 rc $?"
fi
if has stream; then
timeout -k 10 300 python tools/lr_stream_probe.py 2>/dev/null | grep "pass ms" > gpurun_out/${T}_lr_stream_probe.txt; echo "stream rc $?"
fi
if has cpufull; then
timeout -k 10 1100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-adversarial --no-job --sustain-s 0 --cpu-baseline-full > gpurun_out/${T}_c4_bench_cpu_full.json 2> gpurun_out/${T}_c4_bench_cpu_full.err; echo "cpu full rc $?"
fi
