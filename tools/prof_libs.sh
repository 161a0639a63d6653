#!/bin/bash
# Serial kernel stats (rocprofv3 --kernel-trace --stats) of the default path per library build: average duration of the kernels matching a regex.
#   tools/prof_libs.sh "<regex>" "<lib a> <lib b> ..." [extra bench args]      (libraries relative to the repo root)
pat=$1; libs=$2; shift 2
for lib in $libs; do
  export LDW_AMD_LIB="$GRAFT_REPO_ROOT/$lib"
  tag=$(basename "$lib" .so)
  bash "$GRAFT_REPO_ROOT/tools/prof_run.sh" "plib_$tag" --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --no-overlap "$@" > /dev/null 2>&1
  echo "== $lib"
  python3 - "$pat" "$GRAFT_REPO_ROOT/gpurun_out/plib_${tag}_kernel_stats.csv" <<'PY'
import re, sys
pat, f = sys.argv[1], sys.argv[2]
for l in open(f):
    if re.search(pat, l):
        print("  ", l.rstrip()[:230])
PY
done
unset LDW_AMD_LIB
