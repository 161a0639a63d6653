#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <python script> <args...>: rocprofv3 kernel trace + stats of any python tool of this repo,
# condensed summary (top 24 kernels) to gpurun_out/<tag>_kernel_stats.csv
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -o p -- python3 "$GRAFT_REPO_ROOT/$script" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1)
python3 tools/prof_summary.py "$f" 24 > "gpurun_out/${tag}_kernel_stats.csv"
rm -rf "gpurun_out/prof_$tag"
cat "gpurun_out/${tag}_kernel_stats.csv"
