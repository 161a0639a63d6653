#!/bin/bash
# per-block GEMM times (LDW_BLOCK_TRACE) of the serialized replay, tile pruning on and off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in on off; do
  F=""; [ $v = off ] && F="--no-prune"
  LDW_BLOCK_TRACE=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-job $F > gpurun_out/r03tp_$v.json 2> gpurun_out/r03tp_$v.err || exit 1
  grep "ldw block" gpurun_out/r03tp_$v.err | tail -55 > gpurun_out/r03tp_$v.trace
done
paste <(awk '{print $3,$4,$5,$6, $(NF-9)}' gpurun_out/r03tp_on.trace) <(awk '{print $(NF-9), $(NF-7), $(NF-2)}' gpurun_out/r03tp_off.trace) | head -60
