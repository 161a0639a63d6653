#!/bin/bash
# per-dispatch kernel durations of the LAST pass of a short serialized bench run (rocprofv3 --kernel-trace), condensed per block:
# gpurun_out/<tag>_dispatch.txt
tag=${1:-r03dt}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/dt_$tag" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap "$@" > "$GRAFT_REPO_ROOT/gpurun_out/dt_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "gpurun_out/dt_$tag" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "gpurun_out/${tag}_dispatch.txt" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"ldw::(\w+)", n)
    return m.group(1) if m else n[:30]
# split into passes at k_pack_panel bursts is fragile: split into blocks at every gemm_apx_kernel / gemm launch preceded by k_zero4
ev = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# last 55 blocks: find indices of gemm_apx_kernel
idx = [i for i, e in enumerate(ev) if e[0] == "gemm_apx_kernel"]
idx = idx[-55:]
start = idx[0] - 6
blk = []
cur = {}
for i in range(start, len(ev)):
    name, us = ev[i]
    if name == "k_zero4" and cur.get("gemm_apx_kernel") is not None and "k_sel_thresh" in cur or (name == "k_zero4" and "gemm_apx_kernel" in cur and "k_pair_mi" in cur):
        blk.append(cur); cur = {}
    cur[name] = cur.get(name, 0.0) + us
blk.append(cur)
keys = ["gemm_apx_kernel", "k_mi_screen", "k_mi_screen_generic", "k_build_packs", "k_pair_sums", "k_pair_mi", "gemm_bits_kernel", "k_mi_units", "k_sel_thresh"]
print("blk " + " ".join(f"{k[:12]:>12}" for k in keys) + "   other")
for b, c in enumerate(blk):
    oth = sum(v for k, v in c.items() if k not in keys)
    print(f"{b:3d} " + " ".join(f"{c.get(k, 0.0):12.1f}" for k in keys) + f" {oth:8.1f}")
PY
rm -rf "gpurun_out/dt_$tag"
head -70 "gpurun_out/${tag}_dispatch.txt"
