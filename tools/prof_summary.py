#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv: short kernel names, top-N rows.
usage: tools/prof_summary.py <kernel_stats.csv> [N] > profiles/<name>.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
for r in rows[:top]:
    name = r["Name"]
    m = re.search(r"(ldw::\w+(<[\w, ]+>)?)", name)
    if m:
        short = m.group(1)
    else:
        m = re.search(r"rocprim::\w+::detail::(\w+)", name)
        short = ("rocprim::" + re.sub(r"^.*wrapped_(\w+?)_config.*$", r"\1", name)) if "wrapped_" in name else (("rocprim::" + m.group(1)) if m else name[:60])
        if "at::native" in name:
            short = "torch::" + re.sub(r".*at::native::(?:\(anonymous namespace\)::)?(\w+).*", r"\1", name)[:50]
    w.writerow([short, r["Calls"], f"{int(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.1f}", r["Percentage"],
                f"{int(r['MinNs'])/1e3:.1f}", f"{int(r['MaxNs'])/1e3:.1f}"])
