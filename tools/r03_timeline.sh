#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of a few blocks in the middle of the last overlapped pass: gpurun_out/<tag>_timeline.txt
tag=${1:-r03tl}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tl_$tag" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > "$GRAFT_REPO_ROOT/gpurun_out/tl_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "gpurun_out/tl_$tag" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "gpurun_out/${tag}_timeline.txt" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"ldw::(\w+)", n)
    return m.group(1) if m else re.sub(r".*::", "", n)[:28]
g = [i for i, r in enumerate(rows) if "gemm_apx_kernel" in r["Kernel_Name"]]
g = g[-55:]                      # the last pass
i0, i1 = g[20], g[26]            # six blocks in the middle
t0 = int(rows[i0]["Start_Timestamp"])
qs = {}
for r in rows[i0 - 8:i1]:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:10.1f} {d:8.1f}  q{qs[q]}  {'    ' * qs[q]}{short(r['Kernel_Name'])}")
PY
rm -rf "gpurun_out/tl_$tag"
head -150 "gpurun_out/${tag}_timeline.txt"
