#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_s3_gputests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_s3_gputests.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --path 1 > gpurun_out/r02_b3_path1.json 2> gpurun_out/r02_b3_path1.err; echo "bench path1 rc $?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_b3_auto.json 2> gpurun_out/r02_b3_auto.err; echo "bench auto rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r02_b3_path1.json","gpurun_out/r02_b3_auto.json"):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, "ms/step", round(j["ms_per_step"],2), "stages", {k:round(v,1) for k,v in j["stages_ms_per_step"].items()}, "ovl", {k:round(v,1) for k,v in j["stages_ms_per_step_overlapped"].items()}, j["links"], j["counters"])
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/prof_run.sh r02a_c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "prof rc $?"; cat gpurun_out/r02a_c4_kernel_stats.csv
bash tools/prof_run.sh r02a_c4_serial --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1; echo "prof serial rc $?"; cat gpurun_out/r02a_c4_serial_kernel_stats.csv
