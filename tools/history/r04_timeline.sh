#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the TIMED (overlapped) pass of `bench.py --steps 1 --warmup 2`: per-queue busy / idle, kernel totals
# per queue, and the dispatch sequence of a few items in the middle: gpurun_out/<tag>_timeline.txt.  Items per pass are counted from the
# k_pick_bucket* launches (one per item).
tag=${1:-r04tl}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tl_$tag" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" > "$GRAFT_REPO_ROOT/gpurun_out/tl_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "gpurun_out/tl_$tag" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "gpurun_out/${tag}_timeline.txt" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"ldw::(\w+(<[\w, ]+>)?)", n)
    return m.group(1) if m else re.sub(r".*::", "", n)[:28]
picks = [i for i, r in enumerate(rows) if "k_pick_bucket" in r["Kernel_Name"]]
# passes: 2 warm-up + 1 timed (overlapped) + 1 serialized replay; the probes add 2 picks per cold pass
per = len(picks) // 4
seg = picks[2 * per: 3 * per]
lo_i, hi_i = seg[0], seg[-1]
# widen to the whole pass: from the first kernel after the previous pass's last k_block_done
while lo_i > 0 and "k_block_done" not in rows[lo_i - 1]["Kernel_Name"]: lo_i -= 1
while hi_i < len(rows) - 1 and "k_block_done" not in rows[hi_i]["Kernel_Name"]: hi_i += 1
# the last item's selection
nxt = hi_i + 1
span0, span1 = int(rows[lo_i]["Start_Timestamp"]), int(rows[hi_i]["End_Timestamp"])
print(f"# pass: dispatches {hi_i - lo_i + 1}, span {(span1 - span0) / 1e6:.2f} ms, items (picks) {len(seg)}")
busy = {}
for r in rows[lo_i:hi_i + 1]:
    busy.setdefault(r.get("Queue_Id", "?"), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
allv = sorted(x for v in busy.values() for x in v)
u, ce = 0, None
for s_, e_ in allv:
    if ce is None or s_ > ce: u += e_ - s_; ce = e_
    elif e_ > ce: u += e_ - ce; ce = e_
print(f"# union busy (any queue) {u / 1e6:.2f} ms, idle {(span1 - span0 - u) / 1e6:.2f} ms")
for q, iv in busy.items():
    iv.sort()
    tot, gaps, cur_e = 0, [], None
    for s_, e_ in iv:
        tot += e_ - s_
        if cur_e is not None and s_ > cur_e: gaps.append((s_ - cur_e) / 1e3)
        cur_e = e_ if cur_e is None else max(cur_e, e_)
    big = sorted(gaps, reverse=True)[:12]
    print(f"# queue {q}: dispatches {len(iv)}, busy {tot / 1e6:.2f} ms, idle gaps > 30 us: {sum(1 for x in gaps if x > 30)} totalling {sum(x for x in gaps if x > 30) / 1e3:.2f} ms; all gaps {sum(gaps) / 1e3:.2f} ms; largest {[round(x) for x in big]}")
agg = {}
for r in rows[lo_i:hi_i + 1]:
    k = (r.get("Queue_Id", "?"), short(r["Kernel_Name"]))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (q, k), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"# queue {q} {k:34s} calls {n:4d} total {us / 1e3:7.2f} ms  avg {us / n:7.1f} us")
i0 = seg[len(seg) // 2 - 2]
i1 = seg[min(len(seg) - 1, len(seg) // 2 + 2)]
while i0 > 0 and "k_block_done" not in rows[i0 - 1]["Kernel_Name"]: i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
qs = {}
for r in rows[i0:i1 + 8]:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:10.1f} {d:8.1f}  q{qs[q]}  {'    ' * qs[q]}{short(r['Kernel_Name'])}")
PY
rm -rf "gpurun_out/tl_$tag"
head -60 "gpurun_out/${tag}_timeline.txt"
