#!/bin/bash
# r06 second session (split epilogue, new fp64 logarithm): the GPU suite with poisoned allocations (LDW_POISON_ALLOC=2) and a fuzz soak beyond the suite's
# fixed seeds — the plain path (now k_mi_epilogue_fast + _rest) is what every fuzz case compares the other paths with; seed 2303 runs it as the ONE-kernel
# epilogue instead (LDW_NO_EPI_SPLIT).  Output: gpurun_out/r06u_soak.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06u_soak.txt; : > $out
LDW_POISON_ALLOC=2 timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r06u_suite_poison2.log 2>&1
echo "suite under LDW_POISON_ALLOC=2 rc $? : $(tail -1 gpurun_out/r06u_suite_poison2.log)" >> $out
for seed in 2301 2302; do
  timeout -k 10 600 python tools/fuzz_paths.py --cases 100 --seed $seed --mutate mix --extra-weights > gpurun_out/r06u_fuzz_$seed.log 2>&1
  echo "paths mutated+weights seed $seed rc $? $(grep -c ': ok' gpurun_out/r06u_fuzz_$seed.log) ok" >> $out
done
LDW_NO_EPI_SPLIT=1 timeout -k 10 600 python tools/fuzz_paths.py --cases 100 --seed 2303 --mutate mix --extra-weights > gpurun_out/r06u_fuzz_2303.log 2>&1
echo "paths mutated+weights seed 2303 under LDW_NO_EPI_SPLIT rc $? $(grep -c ': ok' gpurun_out/r06u_fuzz_2303.log) ok" >> $out
timeout -k 10 600 python tools/fuzz_sr_model.py --cases 60 --seed 94 > gpurun_out/r06u_fuzzsr_94.log 2>&1
echo "sr_model seed 94 rc $? $(grep -c ': ok' gpurun_out/r06u_fuzzsr_94.log) ok" >> $out
cat $out
