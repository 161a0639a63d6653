#!/bin/bash
# r06 robustness evidence: the GPU suite with every device block filled with in-range garbage at allocation (LDW_POISON_ALLOC=2: words of 1, =3: words of 0xA5A)
# instead of zeroes, and a fuzz soak beyond the suite's fixed seeds.  Output: gpurun_out/r06_soak.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_soak.txt; : > $out
for mode in 2 3; do
  LDW_POISON_ALLOC=$mode timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_suite_poison$mode.log 2>&1
  echo "suite under LDW_POISON_ALLOC=$mode rc $? : $(tail -1 gpurun_out/r06_suite_poison$mode.log)" >> $out
done
for seed in 2101 2102 2103; do
  timeout -k 10 600 python tools/fuzz_paths.py --cases 100 --seed $seed --mutate mix --extra-weights > gpurun_out/r06_fuzz_$seed.log 2>&1
  echo "paths mutated+weights seed $seed rc $? $(grep -c ': ok' gpurun_out/r06_fuzz_$seed.log) ok" >> $out
done
for seed in 91 92; do
  timeout -k 10 600 python tools/fuzz_sr_model.py --cases 60 --seed $seed > gpurun_out/r06_fuzzsr_$seed.log 2>&1
  echo "sr_model seed $seed rc $? $(grep -c ': ok' gpurun_out/r06_fuzzsr_$seed.log) ok" >> $out
done
cat $out
