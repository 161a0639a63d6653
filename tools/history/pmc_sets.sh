#!/bin/bash
# Arbitrary PMC counter sets (one rocprofv3 pass each, kernel-trace only) of the kernels matching a regex over a short serial bench run.
# usage: tools/pmc_sets.sh <regex> <out.json> "<counters of pass 1>" ["<counters of pass 2>" ...]
pat=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_s$i" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-overlap --no-extra-legs > "$GRAFT_REPO_ROOT/gpurun_out/pmc_s$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/pmc_s$i.log"; }
done
cd "$GRAFT_REPO_ROOT"
python3 - "$pat" "$out" <<'PY'
import csv, glob, collections, json, re, sys
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            key = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, d in res.items():
    print(k); print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf gpurun_out/pmc_s[0-9]*
