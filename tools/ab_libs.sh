#!/bin/bash
# A/B of library builds on one box: alternating runs of bench.py (serial stage times from the replay, pass time, link counts) per library and mode.
#   tools/ab_libs.sh "<lib a> <lib b> ..." [reps] [modes: plain default]      (libraries relative to the repo root)
cd "$GRAFT_REPO_ROOT"
libs=$1; reps=${2:-2}; modes=${3:-"plain default"}
for rep in $(seq $reps); do
for lib in $libs; do
  for m in $modes; do
    flags=""; [ "$m" = plain ] && flags="--no-mixed --screen 0 --path 1"
    LDW_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '[$m]', 'ms_per_step', round(d['ms_per_step'], 2), 'serial stages', {k: round(v, 2) for k, v in d['stages_ms_per_step'].items()}, d['links'])"
  done
done
done
