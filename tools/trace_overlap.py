#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: wall span, union of busy intervals, sum of kernel durations, and the
same per kernel family, for the LAST `--steps` worth of dispatches (skips warm-up by taking the trailing fraction).
usage: tools/trace_overlap.py <kernel_trace.csv> [tail_fraction]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in iv)
print(f"dispatches {len(iv)}  wall {(t1 - t0) / 1e6:.2f} ms  union-busy {busy / 1e6:.2f} ms  sum-of-durations {tot / 1e6:.2f} ms  idle {(t1 - t0 - busy) / 1e6:.2f} ms")
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in iv:
    m = re.search(r"(ldw::\w+(<[\w, ]+>)?)", n)
    k = m.group(1) if m else ("rocprim" if "rocprim" in n else n[:40])
    fam[k][0] += e - s; fam[k][1] += 1
for k, (d, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {k:45s} {d / 1e6:9.3f} ms  {c:6d} calls  {d / c / 1e3:8.1f} us avg")
q = collections.Counter(x[3] for x in iv)
print("queues", dict(q))

# per-queue busy time, concurrency, and the idle gaps of every queue
byq = collections.defaultdict(list)
for s, e, n, q in iv:
    byq[q].append((s, e, n))
def union(lst):
    out, cs, ce = [], None, None
    for s, e, *_ in sorted(lst):
        if ce is None or s > ce:
            if ce is not None: out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None: out.append((cs, ce))
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
us = {q: union(l) for q, l in byq.items()}
for q, u in us.items():
    b = sum(e - s for s, e in u)
    gaps = sorted(((u[k + 1][0] - u[k][1]) for k in range(len(u) - 1)), reverse=True)
    print(f"queue {q}: busy {b / 1e6:.2f} ms in {len(u)} stretches; idle gaps: total {sum(gaps) / 1e6:.2f} ms, > 20 us: {sum(1 for g in gaps if g > 20000)} ({sum(g for g in gaps if g > 20000) / 1e6:.2f} ms), largest {[round(g / 1e3) for g in gaps[:5]]} us")
qs = sorted(us, key=lambda q: -sum(e - s for s, e in us[q]))[:2]
if len(qs) == 2:
    print(f"both of {qs} busy: {inter(us[qs[0]], us[qs[1]]) / 1e6:.2f} ms")
if len(sys.argv) > 3:   # timeline of N dispatches from the middle
    n = int(sys.argv[3]); mid = len(iv) // 2
    base = iv[mid][0]
    for s, e, nm, q in iv[mid:mid + n]:
        m = re.search(r"(ldw::\w+(<[\w, ]+>)?)", nm)
        print(f"  q{q} {(s - base) / 1e3:9.1f} -> {(e - base) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f}  {m.group(1) if m else nm[:40]}")
