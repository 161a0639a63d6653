#!/bin/bash
# HBM traffic per launch of every ldw:: kernel from PMC counters, as MI355X_MICROARCH.md prescribes: separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE), kernel-trace only.   usage (on the GPU box): tools/pmc_traffic2.sh <out.json> <bench args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs "$@" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1 || { echo "pmc pass $c failed"; tail -5 "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log"; exit 1; }
done
cd "$GRAFT_REPO_ROOT"
python3 - "$out" "$*" <<'PY'
import csv, glob, collections, json, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"ldw::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            if m and r["Counter_Name"] == c:
                agg[m.group(1)][c].append(float(r["Counter_Value"]))
WIDE = ("gemm_bits_kernel", "gemm_lo_units_kernel", "gemm_mi_fused_kernel")   # 16-B-per-lane coalesced streaming reads
res = {}
for k, d in sorted(agg.items()):
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    raw, dbl = (f + w) * 1024, (2 * f + w) * 1024
    wide = k.startswith(WIDE)
    # `hbm_bytes_per_launch`: the guide's gfx950 correction applied (FETCH_SIZE x 2: it tallies 128-B requests at 64 B for wide coalesced
    # reads).  For kernels whose reads are narrower than 16 B per lane the correction is uncalibrated: the doubled figure is then an UPPER
    # bound and the range [raw, doubled] is what the counters support.
    e = dict(launches=len(d["FETCH_SIZE"]), FETCH_SIZE_KB_mean_per_launch=f, WRITE_SIZE_KB_mean_per_launch=w,
             hbm_bytes_per_launch=dbl, hbm_bytes_per_launch_range=[raw, dbl],
             calibrated=bool(wide),
             note=("16-B-per-lane coalesced reads: FETCH_SIZE x 2 is the guide's calibrated correction" if wide else
                   "reads narrower than 16 B per lane / gathered: uncalibrated on gfx950 — true bytes lie in hbm_bytes_per_launch_range; "
                   "hbm_bytes_per_launch is its upper end"))
    res[k] = e
res["_how"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (another pass) -- python3 bench.py --no-cpu-baseline "
               "--no-extra-legs " + sys.argv[2] + "; means over all launches of each kernel (warm-up, timed and replay steps alike)")
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("launches", "FETCH_SIZE_KB_mean_per_launch", "WRITE_SIZE_KB_mean_per_launch")} for k, v in res.items() if k != "_how"}, indent=0))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
