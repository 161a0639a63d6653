#!/bin/bash
# kernel (+ memory copy) trace of tools/job_profile.py --cold: what runs between the end of the MI pass and the short-range model's first sort
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tt" -o p -- python3 "$GRAFT_REPO_ROOT/tools/job_profile.py" --cold > "$GRAFT_REPO_ROOT/gpurun_out/tt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import csv, glob, re
ev = []
for f in glob.glob("gpurun_out/tt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(ldw::\w+|rocprim::\w+(?:::\w+)*|\w+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:70])))
for f in glob.glob("gpurun_out/tt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s bytes" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
# the first kernel of the quantile step's sort
i0 = next(i for i, e in enumerate(ev) if "rocprim" in e[2] or "SrPay" in e[2] or "radix" in e[2].lower())
t0 = ev[i0][0]
print("# events in the 120 ms before the first sort kernel (ms relative to it, duration ms)")
for s, e, n in ev:
    if t0 - 120e6 <= s <= t0 + 5e6 and (e - s > 200e3 or s > t0 - 2e6 or "C " in n[:2]):
        print("%9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
PY
rm -rf gpurun_out/tt
