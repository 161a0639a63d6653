#!/bin/bash
# the driver's N > 1 command with 2 and 4 ranks on the ONE GPU of the box (backend gloo: transfers go through the host, the ranks
# share the GPU — the times are not RCCL times; what this shows is the per-rank record and the host-side overheads of a phase)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29$((500+n)) bench.py --gpus $n --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/r03_ranks${n}_bench.json 2> gpurun_out/r03_ranks${n}_bench.err; echo "n=$n rc $?"
  python3 - $n <<'PY'
import json,sys
j=json.loads([l for l in open(f"gpurun_out/r03_ranks{sys.argv[1]}_bench.json") if l.startswith("{")][0])
print({k: j[k] for k in ("n_gpus","ms_per_step","value","links")})
for r in j["per_rank"]: print({k:(round(v,2) if isinstance(v,float) else v) for k,v in r.items()})
PY
done
