#!/bin/bash
# SQ counters of the kernels whose name matches $1 (regex), one rocprofv3 --pmc pass over a short bench run.
# usage (on the GPU box): tools/pmc_kernel.sh <regex> <out.json> [bench args]
pat=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_k" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-overlap "$@" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_k.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$pat" "$out" <<'PY'
import csv, glob, collections, json, re, sys
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_k/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            key = re.sub(r"\(.*", "", r["Kernel_Name"])[:90] + " grid=" + r.get("Grid_Size", "?") + " vgpr=" + r.get("VGPR_Count", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, d in res.items():
    print(k); print("   ", {c: (round(v) if v > 10 else v) for c, v in d.items()})
PY
rm -rf gpurun_out/pmc_k
