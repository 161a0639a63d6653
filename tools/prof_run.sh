#!/bin/bash
# usage (on the GPU box): tools/prof_run.sh <tag> <bench.py args...>: rocprofv3 kernel trace + stats of one bench.py run,
# condensed summary to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1)
python3 tools/prof_summary.py "$f" 14 > "gpurun_out/${tag}_kernel_stats.csv"
grep '^{' "gpurun_out/prof_$tag.log" > "gpurun_out/${tag}_bench.json"
rm -rf "gpurun_out/prof_$tag"
cat "gpurun_out/${tag}_kernel_stats.csv"
