#!/bin/bash
# r06, first evidence run: whole GPU suite (durations), the driver's bench command, the Hamming probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r06_gputest.log 2>&1; echo "tests rc $?"; tail -22 gpurun_out/r06_gputest.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c4_driver_bench.json 2> gpurun_out/r06_c4_driver_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_c4_driver_bench.json") if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "mi_produced ms", d.get("ms_per_step_mi_produced"))
print("roofline_hamming", json.dumps(d.get("roofline_hamming"))[:900])
sm=d.get("scaling_model",{}).get("predicted",{})
for n,v in sm.items(): print("N",n,{k:(round(x,2) if isinstance(x,float) else x) for k,x in v.items() if k!="per_rank"}, [round(r["compute_ms"],1) for r in v["per_rank"]])
print("epilogue", json.dumps(d.get("roofline_mi_produced",{}).get("epilogue"))[:700])
PY
