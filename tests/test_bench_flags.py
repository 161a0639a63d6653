"""GPU: bench.py under every switch combination that changes the code path (tools/flag_matrix.py) on a small workload: each
run exits 0 and all runs report the same link counts — the switches change HOW the links are found, never WHICH.  The big
shapes (85k x 616, 500k x 10k) run through the same script by hand; their records are kept under profiles/."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import flag_matrix  # noqa: E402

pytestmark = pytest.mark.gpu


def test_bench_flag_matrix_small():
    out, bad, counts = flag_matrix.run_matrix("--L 12000 --N 1200 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs", timeout=600, log=lambda *_: None)
    assert not bad, [(r["flags"], r.get("stderr_tail", "")[-600:]) for r in bad]
    assert len(counts) == 1 and all(c is not None and c > 0 for c in next(iter(counts))), counts
    assert len(out) == len(flag_matrix.COMBOS)
