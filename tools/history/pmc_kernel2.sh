#!/bin/bash
# SQ counters (two passes) of kernels matching $1 over a short bench run.  usage: tools/pmc_kernel2.sh <regex> <out.json> [bench args]
pat=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_k$i" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-overlap "$@" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_k$i.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
python3 - "$pat" "$out" <<'PY'
import csv, glob, collections, json, re, sys
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_k*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            key = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, d in res.items():
    print(k); print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf gpurun_out/pmc_k1 gpurun_out/pmc_k2
