#!/usr/bin/env python3
"""bench.py under every switch combination that changes the code path: each run must exit 0 and all runs of one
workload must report the same link counts (the switches change HOW the links are found, never WHICH).

    python tools/flag_matrix.py [--L 30000] [--N 2000] [--extra "--L 85000 --N 616"] [--out gpurun_out/flags.json]

Used by tests/test_bench_flags.py (small workload) and by hand for the big shapes (C3-like 85k x 616, C5 500k x 10k).
"""
import argparse
import json
import os
import shlex
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _experiments() -> bool:
    sys.path.insert(0, ROOT)
    from ldweaver_amd import _lib
    return _lib.has_experiments()


COMBOS = [
    "",
    "--no-mixed",
    "--screen 0",
    "--screen 2",
    "--no-mixed --screen 0",
] + (["--fused"] if _experiments() else []) + [   # (the fused kernel is not in the default library: r05)
    "--no-overlap",
    "--no-mixed --no-overlap",
    "--path 1",
    "--path 2",
    "--path 2 --screen 2",
    "--path 2 --no-overlap",
    "--engine hist",
    "--warm",
]


def run_one(base: str, flags: str, timeout: int):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + shlex.split(base) + shlex.split(flags)
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
    rec = dict(flags=flags, rc=p.returncode, wall_s=round(time.time() - t0, 1))
    if line:
        j = json.loads(line)
        rec.update(n_sr=j["links"].get("n_sr"), n_lr=j["links"].get("n_lr"), ms_per_step=j["ms_per_step"], counters=j.get("counters"))
    else:
        rec["stderr_tail"] = p.stderr[-1500:]
    return rec


def run_matrix(base: str, combos=COMBOS, timeout: int = 900, log=print):
    out = []
    for f in combos:
        rec = run_one(base, f, timeout)
        log(json.dumps(rec))
        out.append(rec)
    bad = [r for r in out if r["rc"] != 0]
    counts = {(r.get("n_sr"), r.get("n_lr")) for r in out if r["rc"] == 0}
    return out, bad, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default="--L 30000 --N 2000 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs")
    ap.add_argument("--only", default=None, help="comma-separated indices into the combination list")
    ap.add_argument("--out", default=None)
    ap.add_argument("--timeout", type=int, default=900)
    a = ap.parse_args()
    combos = COMBOS if a.only is None else [COMBOS[int(i)] for i in a.only.split(",")]
    out, bad, counts = run_matrix(a.base, combos, a.timeout)
    res = dict(base=a.base, runs=out, failed=len(bad), distinct_link_counts=sorted(map(list, counts)))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(dict(failed=len(bad), distinct_link_counts=res["distinct_link_counts"])))
    sys.exit(1 if bad or len(counts) != 1 else 0)


if __name__ == "__main__":
    main()
