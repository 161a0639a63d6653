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
