#!/bin/bash
# round 6, second session: evidence for the split epilogue — SQ counters of the epilogue kernels, the GPU suite, the driver's bench line, kernel stats
# of the plain path (serial).  Everything lands in gpurun_out/ under the tag r06s (copy what is quoted into profiles/).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/pmc_epilogue.sh gpurun_out/r06s_pmc_epilogue.json; echo "pmc rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r06s_gputest.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r06s_gputest.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06s_c4_driver_bench.json 2> gpurun_out/r06s_c4_driver_bench.err; echo "bench rc $?"
bash tools/prof_run.sh r06s_c4_plain_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap --no-mixed --screen 0 --path 1 > /dev/null 2>&1; echo "prof plain rc $?"
bash tools/prof_run.sh r06s_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"
