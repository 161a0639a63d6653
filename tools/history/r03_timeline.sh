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
g = g[-165:-110]                 # the last TIMED (overlapped) pass: bench.py ends with two serialized replay passes
i0, i1 = g[20], g[26]            # six blocks in the middle
sys.stderr.write("columns: " + ",".join(rows[0].keys()) + "\n")
t0 = int(rows[i0]["Start_Timestamp"])
# busy / idle time of every queue over the whole pass
lo_i, hi_i = g[0] - 6, min(len(rows), g[-1] + 30)
span0, span1 = int(rows[lo_i]["Start_Timestamp"]), int(rows[hi_i - 1]["End_Timestamp"])
busy = {}
for r in rows[lo_i:hi_i]:
    q = r.get("Queue_Id", "?")
    busy.setdefault(q, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, iv in busy.items():
    iv.sort()
    tot, gaps, cur_e = 0, [], None
    for s_, e_ in iv:
        tot += e_ - s_
        if cur_e is not None and s_ > cur_e: gaps.append((s_ - cur_e) / 1e3)
        cur_e = e_ if cur_e is None else max(cur_e, e_)
    big = sorted(gaps, reverse=True)[:12]
    print(f"# queue {q}: pass span {(span1 - span0) / 1e6:.2f} ms, busy {tot / 1e6:.2f} ms, idle gaps > 30 us: {sum(1 for x in gaps if x > 30)} totalling {sum(x for x in gaps if x > 30) / 1e3:.2f} ms; largest {[round(x) for x in big]}")
agg = {}
for r in rows[lo_i:hi_i]:
    k = (r.get("Queue_Id", "?"), short(r["Kernel_Name"]))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (q, k), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"# queue {q} {k:28s} calls {n:4d} total {us / 1e3:7.2f} ms  avg {us / n:7.1f} us")
qs = {}
for r in rows[i0 - 8:i1]:
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    qs.setdefault(q, len(qs))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:10.1f} {d:8.1f}  q{qs[q]}  {'    ' * qs[q]}{short(r['Kernel_Name'])}")
PY
rm -rf "gpurun_out/tl_$tag"
head -150 "gpurun_out/${tag}_timeline.txt"
