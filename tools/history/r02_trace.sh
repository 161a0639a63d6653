#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/trc" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline $BENCH_FLAGS > "$GRAFT_REPO_ROOT/gpurun_out/trc.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/trc -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$f" 0.3 ${DETAIL:-0}
grep -o '"ms_per_step": [0-9.]*' gpurun_out/trc.log | head -1
rm -rf gpurun_out/trc
