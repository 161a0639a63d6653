#!/bin/bash
# quick loop: apx tests, bench (auto path), serial profile
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "apx or multiallelic" > gpurun_out/r02_q_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_q_tests.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_FLAGS > gpurun_out/r02_q_bench.json 2> gpurun_out/r02_q_bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r02_q_bench.json",):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, "ms/step", round(j["ms_per_step"],2), "stages", {k:round(v,1) for k,v in j["stages_ms_per_step"].items()}, "ovl", {k:round(v,1) for k,v in j["stages_ms_per_step_overlapped"].items()}, j["links"], j["counters"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r02_q_bench.err").read()[-2000:])
PY
bash tools/prof_run.sh r02q_serial --steps 3 --warmup 1 --no-cpu-baseline --no-overlap $BENCH_FLAGS > /dev/null 2>&1; echo "prof serial rc $?"; head -24 gpurun_out/r02q_serial_kernel_stats.csv
