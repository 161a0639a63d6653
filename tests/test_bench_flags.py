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


def _bench_line(cmd, timeout):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LDW_BENCH_SELF_LAUNCHED")}   # (no launcher around the child)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_rank_rehearsal():
    """`python bench.py --gpus 4` (self-launching) = the command the driver's scaling run uses (torch.distributed.run, one rank per process, phased gather to rank 0), rehearsed on
    the ONE GPU of the test box with backend gloo: 4 ranks — the box's process guard allows at most 6 processes on its card, and the
    launcher and this pytest process may count among them (a 6-rank run was killed by it), so the 8-rank deal + gather is covered
    on the CPU instead (tests/test_dist_gloo.py::test_gather_gloo[8]).  Exit code 0, link counts
    identical to the 1-rank line, and the per-rank record the N > 1 line carries (compute / exposed gather / bytes sent)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = ["--steps", "1", "--warmup", "1", "--no-extra-legs", "--no-cpu-baseline", "--L", "40000", "--N", "2000", "--max-blk-sz", "5000"]
    one = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + base, 600)
    world = 4
    # r04: no launcher around it — `python bench.py --gpus 4` starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node 4
    # --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` itself (a child process, before this one touches the GPU),
    # relays rank 0's line and the exit code
    # (the 4-rank line runs its extra leg: `sr_tail`, the short-range model + ARACNE behind the pass with the table gathered / the rows left on their ranks)
    # (--min-blocks-per-phase 3: the rehearsal keeps the three-phase gather in it — at 9 block pairs per rank the default, 12, would run one phase)
    many = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--min-blocks-per-phase", "3"] + [a for a in base if a != "--no-extra-legs"], 900)
    assert many["ranks_seen"] == world and many["backend"] == "gloo" and many["self_launched"] is True
    assert one["ranks_seen"] == 1 and one["self_launched"] is False
    assert many["n_gpus"] == world and one["n_gpus"] == 1
    assert many["links"] == one["links"] and one["links"]["n_lr"] > 0 and one["links"]["n_sr"] > 0
    assert many["config"]["pairs"] == one["config"]["pairs"]
    pr = many["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world)) and sum(r["blocks"] for r in pr) == 36
    assert all(r["compute_ms"] > 0 and r["exposed_gather_ms"] >= 0 and r["phases"] == 3 for r in pr)
    assert pr[0]["bytes_sent"] == 0 and all(r["bytes_sent"] > 0 for r in pr[1:])
    # every rank but 0 sends exactly its rows: r04: 8 bytes per short-range row (the MI column; rank 0 rebuilds the index columns from the
    # positions), 16 per long-range row — about half of r03's 16 bytes per row
    # r05: the same job with the short-range rows left on their ranks — same kept links bit for bit, a fraction of the bytes
    tail = many["sr_tail"]
    assert tail["kept_links_equal"] is True and tail["kept_links"]["dist"]["rows"] > 100, tail["kept_links"]
    assert tail["max_bytes_sent_per_peer"]["dist"] < 0.4 * tail["max_bytes_sent_per_peer"]["gather"], tail["max_bytes_sent_per_peer"]
    assert all(r["candidates"] < 0.25 * r["sr_rows"] for r in tail["per_rank"] if r["sr_rows"] > 0)
    # ... and the leg can never cost the line: with a watchdog of 50 ms every rank abandons it, rank 0 prints the line without it, exit code 0
    cut = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--sr-tail-timeout", "0.05"] + [a for a in base if a != "--no-extra-legs"], 600)
    assert cut["n_gpus"] == 2 and cut["links"] == one["links"] and "abandoned" in cut["sr_tail"]["note"]
    sent = sum(r["bytes_sent"] for r in pr)
    assert sent <= 8 * one["links"]["n_sr"] + 16 * one["links"]["n_lr"]
    assert sent < 0.6 * 16 * (one["links"]["n_sr"] + one["links"]["n_lr"]) * 3 / 4 + 16 * one["links"]["n_lr"]


def test_bench_line_carries_roofline_hamming_and_the_scaling_model():
    """r06: `roofline_hamming` (the Hamming stage priced: its GEMM against the int8 peak on executed operations, the kernels around it against HBM, the
    rest of the call as host time) and `scaling_model` (every share of the N = 1, 2, 4, 8 deals run alone on this GPU + a modelled gather; marked as a model)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--L", "30000", "--N", "1500", "--max-blk-sz", "5000", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline", "--no-adversarial", "--no-job", "--sustain-s", "0"], 600)
    rh = d["roofline_hamming"]
    assert rh["columns"] > 30000 and 0 < rh["gemm"]["frac"] < 1 and rh["gemm"]["executed_ops"] > 0 and rh["gemm"]["bound"] == "mfma"
    assert rh["pre"]["bound"] == "hbm" and 0 < rh["pre"]["frac"] < 1 and 0 < rh["post"]["frac"] < 1
    assert abs(rh["kernels_ms"] + rh["host_ms"] - rh["wall_ms"]) < 1e-6 and rh["host_ms"] > 0
    sm = d["scaling_model"]
    assert sm["status"].startswith("MODEL, UNMEASURED")
    pred = sm["predicted"]
    assert sorted(pred) == ["1", "2", "4", "8"]
    assert all(p["predicted_ms_per_step"] > 0 and len(p["per_rank"]) == int(n) for n, p in pred.items())
    assert sum(r["blocks"] for r in pred["8"]["per_rank"]) == 21     # make_blocks(30000, 5000): 6 from-blocks -> 21 block pairs, every one dealt once
    assert pred["1"]["exposed_gather_model_ms"] == 0 and pred["8"]["exposed_gather_model_ms"] > 0
    assert 0.5 < pred["1"]["predicted_ms_per_step"] / d["ms_per_step"] < 2.0
    assert pred["8"]["slowest_share_compute_ms"] < pred["1"]["slowest_share_compute_ms"]
    ep = d["roofline_mi_produced"]["epilogue"]
    assert ep["kernel"].startswith("k_mi_epilogue_fast") and ep["ps_per_pair"] > 0 and ep["valu_issue_frac"] is None   # (the counters are those of the 100k x 5k shape only)
