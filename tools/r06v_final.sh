#!/bin/bash
# the last evidence of round 6 at the committed code (tag r06v: after the pair-sums grids and the host / device logarithm header): GPU suite, the driver's bench line,
# serial kernel stats of the default path
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r06v_gputest.log 2>&1; echo "tests rc $?"; tail -8 gpurun_out/r06v_gputest.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06v_c4_driver_bench.json 2> gpurun_out/r06v_c4_driver_bench.err; echo "bench rc $?"
bash tools/prof_run.sh r06v_c4_serial --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"
bash tools/prof_run.sh r06v_c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; echo "prof rc $?"
