#!/usr/bin/env python3
"""Headline benchmark: MI SNP-pairs/sec of the all-pairs weighted-MI path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--L 100000] [--N 5000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over the whole synthetic alignment: every block pair of
make_blocks(L, 10000) -> MI of every SNP pair -> short-range table + per-block top long-range links, final link
tables assembled on rank 0 (BASELINE.json: synthetic 100k SNPs x 5k sequences, reference defaults).  The state
matrix, weights and SNP meta data are resident in HBM before the timed region.  With N > 1 the block pairs are
dealt over the ranks (total work fixed -> "strong" scaling) and gathered with one RCCL gatherv.
Prints ONE JSON line on rank 0.

Every step is a COLD pass, as a job runs it: `ldw_reset_speculation` forgets the histogram-bucket guesses, their spread history
and the biallelic threshold table before each step (warm-up steps included), so a step visits every block pair once without
anything learnt from an earlier pass over the same data (R/computePairwiseMI.R:103-116).  Buffers stay allocated.

Order of work (the GPU section is one stretch, the CPU baseline follows it: r04 — a GPU left idle for the baseline's 20-30 s made the first pass
after it 30 ms slower than the first pass of a job, which follows the set-up at once): setup ->
W warm-up steps -> EXACTLY K timed steps -> (N = 1 only) the extra
legs reported on the same line: `warm_replay` (K steps WITHOUT the reset: every pass inherits its predecessor's guesses — the
r02 headline, kept for comparison), `sustained` (cold steps until >= 10 s of GPU work have run), the kernel-exclusive replay
behind `roofline`, `mi_values_produced` (5-limb GEMM + fp64 MI of EVERY pair: no screen, no mixed precision, no approximate
GEMM — the rate at which MI values, not decisions, are produced) and `job` (the product entry points end to end, state matrix
on the HOST -> both tsv files on disk: `h2d_ms`, Hamming weights, cold MI pass, `tsv_write_s`, short-range model + ARACNE), the
`adversarial` workload; last, `cpu_baseline` on the host cores (rank 0, N = 1 only).
With N > 1 the line carries `per_rank` (compute_ms, exposed_gather_ms, bytes_sent of every rank).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ldweaver_amd.cpushare import cpu_share, limit_thread_pools  # noqa: E402  (before numpy: see the module)

# (one process per GPU: every rank of a node takes its part of the cgroup's CPU share)
limit_thread_pools(max(1, cpu_share() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=100_000)
    ap.add_argument("--N", type=int, default=5_000)
    ap.add_argument("--max-blk-sz", type=int, default=10_000)
    ap.add_argument("--nlimbs", type=int, default=0)
    ap.add_argument("--engine", choices=["mfma", "hist", "hist_states"], default="mfma",
                    help="mfma: the default path; hist: joint histograms on bit planes (class-wise popcounts, k_cooc_popc) + the same fp64 epilogue; "
                         "hist_states: the first histogram kernel (byte states in LDS, k_mi_hist)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for "
                                                      "smoke-testing the N > 1 path on a single GPU)")
    ap.add_argument("--no-prune", action="store_true", help="A/B: no row ordering / tile pruning in the approximate GEMM")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the timed region with the GEMM/epilogue stream overlap off too (kernel-exclusive times everywhere; "
                         "the command profiles/*_serial_kernel_stats.csv was collected with)")
    ap.add_argument("--fused", action="store_true", help="GEMM + MI epilogue as one kernel (ldw_set_fused(1)); default is GEMM -> G -> "
                                                         "k_mi_screen -> k_mi_units, which is faster (docs/HISTORY.md 5.2)")
    ap.add_argument("--no-mixed", action="store_true", help="block-wide GEMM with all 5 limbs instead of 3 high limbs + gathered low limbs")
    ap.add_argument("--path", type=int, default=0, help="block-wide pass of the speculative blocks: 0 auto, 1 limb GEMM paths, 2 approximate GEMM + popcount sums")
    ap.add_argument("--screen", type=int, default=1, help="fp32 screen before the fp64 MI evaluation: 0 off, 1 on, 2 verify")
    ap.add_argument("--gather-phases", type=int, default=3, help="N > 1: phases of the link-table gather (1 = one gather after all blocks)")
    ap.add_argument("--min-blocks-per-phase", type=int, default=12, help="N > 1: a phase of the gather holds at least this many block pairs of a rank's share (a phase is its own "
                                                                            "ldw_mi_all_pairs call: ~0.65 ms + a shorter span plan; measured in profiles/r06_gather_phases_model.txt)")
    ap.add_argument("--full-sr-rows", dest="sr_mi_only", action="store_false", help="N > 1: send all three columns of the short-range rows (r03) instead of "
                                                                                   "their MI column alone (rank 0 rebuilds the index columns: 8 instead of 16 bytes per row)")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="CPU baseline on the sample BASELINE.md 3 states: one diagonal + one off-diagonal 10 000 x 10 000 "
                                                                     "block at the config's N (minutes on 16 cores; the default times a 2 000 x 2 000 sub-block pair)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="SNPs per side of the CPU-baseline sample block (0 = auto)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the warm / sustained / mi_values_produced / job legs after the timed region")
    ap.add_argument("--warm", action="store_true", help="do NOT reset the speculation state before every step (the r02 behaviour: each step inherits "
                                                        "the previous step's bucket guesses); the default is a cold pass per step")
    ap.add_argument("--no-job", action="store_true", help="skip the end-to-end job leg (host states -> tsv files)")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the adversarial-data leg (MAF uniform in [0.2, 0.5], no clonal groups)")
    ap.add_argument("--inproc", action="store_true", help="N > 1 WITHOUT one process per GPU: this one process creates one context per device and runs "
                                                          "ldw_mi_all_pairs_multi (worker threads + peer-to-peer gather inside the library) — the route the R shim takes")
    ap.add_argument("--inproc-devices", default="", help="comma-separated device ids for --inproc (default 0..gpus-1; '0,0' = two contexts on one GPU)")
    ap.add_argument("--no-sr-tail-leg", action="store_true", help="N > 1: skip the leg that runs the short-range model + ARACNE behind the pass both ways "
                    "(table gathered to rank 0 / rows left on their ranks: ldweaver_amd/dist_srp.py)")
    ap.add_argument("--sr-tail-timeout", type=float, default=240.0, help="N > 1: seconds the sr_tail leg may take before every rank abandons it and rank 0 prints the line without it")
    ap.add_argument("--strict-exit", action="store_true", help="N > 1: exit code 3 when the sr_tail leg is abandoned or fails AFTER rank 0 has printed the line (default 0: the line of "
                                                                "the timed region is complete and says in `sr_tail.note` what happened to the leg; a launcher that drops the line of a non-zero exit would lose the measurement)")
    ap.add_argument("--sustain-s", type=float, default=10.0, help="N = 1: keep stepping after the timed region until this many seconds of steps have run")
    return ap.parse_args()


def cpu_baseline(states_np, hdw, r, uqe, N, sample):
    """The C/OpenMP block-faithful restatement (oracle/ldw_oracle.c, kind 'port') on one diagonal and one
    off-diagonal sample block of the same workload, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    cores = min(c_oracle.max_threads(), cpu_share())   # (the cgroup's share, not the host's 256 CPUs: more threads than that only get the process throttled)
    s = sample
    fi, ti = np.arange(0, s), np.arange(s, 2 * s)
    t0 = time.time()
    c_oracle.mi_block(states_np, hdw, r, uqe, fi, fi, cores)
    c_oracle.mi_block(states_np, hdw, r, uqe, fi, ti, cores)
    dt = time.time() - t0
    pairs = s * (s - 1) // 2 + s * s
    stated = s >= 10_000   # BASELINE.md 3 / SURVEY 8(d): one diagonal + one off-diagonal block of max_blk_sz = 10 000 SNPs at the config's N
    return dict(value=pairs / dt, unit="MI SNP-pairs/s", cores=cores, kind="port", sample_is_stated_block=bool(stated), sample_pairs=int(pairs), sample_s=dt,
                sample=(f"one diagonal + one off-diagonal {s}x{s} block of the workload (its first two 10000-SNP blocks) at N={N}: the sample BASELINE.md 3 states "
                        f"(all 25 state pairs, dense x CSR + fused Hadamard), {dt:.1f} s wall" if stated else
                        f"one diagonal + one off-diagonal {s}x{s} sub-block of the workload's first 10000-SNP block at N={N} (all 25 state "
                        f"pairs, dense x CSR + fused Hadamard), {dt:.1f} s wall.  NOT the stated sample (a whole 10000 x 10000 block pair is 25x the pairs: "
                        f"minutes on these cores — `--cpu-baseline-full` runs it); the cost is linear in pairs at fixed N, so the sample rate extrapolates"))


def phases_for(nblocks, world, max_phases, min_blocks=12):
    """Phases of the gather for a rank's share.  A phase is its own ldw_mi_all_pairs call: ~0.65 ms of fixed cost, its own span plan over fewer blocks.  Measured
    on one GPU (every share run alone, profiles/r06_gather_phases_model.txt): slowest share at N = 2 / 4 / 8 with 1 phase 20.0 / 11.2 / 6.8 ms, with 3 (2 at N = 8)
    phases 22.0 / 13.4 / 8.4 ms — more than the transfer a phase hides at 153 GB/s per link.  r06: a phase holds at least NINE block pairs (r03-r05: three), so at
    config 4 only N = 2 is phased (two phases: predicted 23.5 ms against 23.8 with one and 25.1 with three)."""
    return 1 if world == 1 else max(1, min(max_phases, (nblocks // world) // max(1, min_blocks)))


def scaling_model_leg(eng, blocks, sr_dist, lr_retain, approx, pairs, gather_phases, measured_n, measured_ms, min_blocks=12):
    """a PREDICTED strong-scaling curve on every line, so that the first multi-GPU record has a stated expectation to be held against.
    No number here was measured on more than one GPU.  What is measured, on THIS GPU: for N = 1, 2, 4, 8 the cost-weighted deal of the block pairs
    (dist.deal_blocks, the deal the N-rank run makes) and every rank's share run ALONE as the N-rank run runs it — a cold start (probes), its phases as
    separate ldw_mi_all_pairs calls — so call overheads, per-rank probes and the shorter spans of a small share are in `compute_ms`.  What is modelled:
    the gather.  Rows of a finished phase travel while the next is computed (dist.gather_begin), so only the LAST phase's transfer is exposed: its
    bytes (8 B per short-range row: the MI column; 16 B per long-range row) over the peer's own xGMI link to rank 0 at 153 GB/s (MI355X_MICROARCH.md: 7
    links per GPU, point to point), then rank 0 re-interleaves the segments into make_blocks order (read + write of the assembled table at 3 TB/s),
    rebuilds the short-range index columns (ldw_sr_pairs_fill: 0.76 ms at this shape, profiles/r05_sr_pairs_probe.txt) and pays ~0.15 ms per phase
    for the count all-reduce + grouped send / receive launches.  predicted_ms = slowest share + exposed gather."""
    import torch
    from ldweaver_amd.dist import deal_blocks
    link_GBps, hbm_copy_GBps, fill_ms, coll_ms = 153.0, 3000.0, 0.76, 0.15
    nblocks = len(blocks)
    pred = {}
    for n in (1, 2, 4, 8):
        shares = deal_blocks(blocks, n)
        n_phase = phases_for(nblocks, n, gather_phases, min_blocks)
        ranks = []
        for rk, mine in enumerate(shares):
            phases = np.array_split(mine, n_phase)
            best, rows = None, None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.reset_speculation()
                rows = []
                for sub in phases:
                    if len(sub):
                        eng.mi_all_pairs(blocks[sub], sr_dist, lr_retain, approx)
                        st = eng.block_stats()
                        rows.append(8 * int(np.sum(st["n_sr"])) + 16 * int(np.sum(st["n_lr_kept"])))
                    else:
                        rows.append(0)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                best = ms if best is None else min(best, ms)
            ranks.append(dict(rank=rk, blocks=int(len(mine)), compute_ms=best, table_bytes=int(sum(rows)), last_phase_bytes=int(rows[-1])))
        compute = max(r["compute_ms"] for r in ranks)
        total_bytes = sum(r["table_bytes"] for r in ranks)
        if n == 1:
            gather = 0.0
        else:
            gather = (max(r["last_phase_bytes"] for r in ranks[1:]) / (link_GBps * 1e9) * 1e3 + 2.0 * total_bytes / (hbm_copy_GBps * 1e9) * 1e3 + fill_ms + coll_ms * n_phase)
        pred[str(n)] = dict(predicted_ms_per_step=compute + gather, predicted_value=pairs / ((compute + gather) * 1e-3), slowest_share_compute_ms=compute,
                            exposed_gather_model_ms=gather, phases=n_phase, per_rank=ranks)
    return dict(status="MODEL, UNMEASURED: no run of this repository has seen more than one GPU (SCALE_r01..r05 were skipped by the driver)",
                predicted=pred, measured_here=dict(n_gpus=measured_n, ms_per_step=measured_ms),
                assumptions=dict(xgmi_link_GBps=link_GBps, rank0_reinterleave_GBps=hbm_copy_GBps, sr_pairs_fill_ms=fill_ms, collective_launch_ms_per_phase=coll_ms,
                                 exposed="last phase's bytes of the slowest peer; earlier phases overlap the next phase's compute",
                                 not_modelled="RCCL start-up (outside the timed region), contention of 7 inbound links at rank 0's HBM (1.07 TB/s of 8), clock differences between GPUs"),
                how="every share of dist.deal_blocks(blocks, N) run alone on this GPU, cold, in its phases (best of 2); transfer from bytes / link rate")


def epilogue_roofline(ms_per_step, pairs, shape):
    """The plain path's fp64 epilogue (one MI per pair; r06: k_mi_epilogue_fast for the straight-line units + k_mi_epilogue_rest for the listed rest) against
    the VALU issue port: SQ counters of the same command, collected in their own rocprofv3 --pmc passes (tools/pmc_epilogue.sh ->
    profiles/r06s_pmc_epilogue.json; the bench cannot run under the counters itself).  A wave64 VALU instruction holds its SIMD's issue port for 4 cycles
    and a CU has 4 SIMDs: one VALU instruction per CU-cycle is the roof; valu_issue_frac = SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES, valu_busy_frac counts the
    quarter-rate v_rcp_f64's extra cycles (SQ_ACTIVE_INST_VALU)."""
    out = dict(kernel="k_mi_epilogue_fast + k_mi_epilogue_rest", bound="valu issue (fp64)", ms_per_step=ms_per_step, ps_per_pair=ms_per_step * 1e-3 / pairs * 1e12,
               valu_issue_frac=None)
    path = os.path.join(ROOT, "profiles", "r06s_pmc_epilogue.json")
    if os.path.exists(path) and shape == (100_000, 5_000, 1):
        allk = json.load(open(path))
        fast = [(k, e) for k, e in allk.items() if isinstance(e, dict) and "k_mi_epilogue_fast" in k and "derived" in e]
        if fast:
            k, ent = max(fast, key=lambda ke: ke[1]["SQ_INSTS_VALU"] * ke[1]["launches"])   # (the instantiation the pass runs: reference quirk, speculative candidates)
            d = ent["derived"]
            out.update(valu_issue_frac=d["valu_issue_frac"], valu_busy_frac=d["valu_busy_frac"], counters_kernel=k.replace("void ", ""),
                       valu_insts_per_pair=ent["SQ_INSTS_VALU"] * 64.0 * 55 / pairs, salu_per_valu=d["salu_per_valu"],
                       wait_inst_any_frac=d["wait_inst_any_frac"], waves_per_simd=5,
                       counters_source="profiles/r06s_pmc_epilogue.json (per-launch averages of the plain path's launches; separate --pmc passes)",
                       note="~4.6 cells per pair x (1 v_rcp_f64 + ~27 fp64 / integer operations) + ~45 per pair of addressing, loads and emission; "
                            "r05's one-kernel epilogue: 268 VALU instructions per pair at 0.655 of the port (3 waves per SIMD, 156 VGPRs, 102 spilled SGPRs)")
    return out


def hamming_roofline(hs, gs, wall_s, L, N):
    """the Hamming stage priced like the MI pass.  Its GEMM (`gemm_bits_kernel<1>` over ~1.3 L bit columns, tiles on or below the
    diagonal) against the dense int8 peak on the operations it EXECUTES (ldw_gemm_stats); the kernels around it (column bits, bit transpose, per-sequence
    counts in front; the N x N neighbour count behind) against HBM on their algorithmic bytes; what is left of the call's wall clock is host work
    (state counts to the host, the column list, allocations, uploads).  SURVEY 8(d): ops_alg = N^2 / 2 x L state compares."""
    out = dict(columns=hs["columns"], wall_ms=wall_s * 1e3, kernels_ms=hs["pre_ms"] + hs["gemm_ms"] + hs["post_ms"],
               host_ms=wall_s * 1e3 - (hs["pre_ms"] + hs["gemm_ms"] + hs["post_ms"]))
    if hs["gemm_ms"] > 0 and gs["bits_ops"] > 0:
        ach = gs["bits_ops"] / (hs["gemm_ms"] * 1e-3) / 1e12
        out["gemm"] = dict(kernel="gemm_bits_kernel<1>", bound="mfma", ms=hs["gemm_ms"], executed_ops=gs["bits_ops"], achieved=ach, peak=5000.0, unit="TFLOP/s", frac=ach / 5000.0,
                           alg_ops=0.5 * N * N * L, alg_frac=0.5 * N * N * L / (hs["gemm_ms"] * 1e-3) / 1e12 / 5000.0)
    for key, ms, nbytes, what in (("pre", hs["pre_ms"], hs["pre_bytes"], "k_hamming_cols + k_bits_transpose + k_seq_minor_count"), ("post", hs["post_ms"], hs["post_bytes"], "k_hdw")):
        if ms > 0:
            out[key] = dict(kernels=what, bound="hbm", ms=ms, alg_bytes=nbytes, achieved=nbytes / (ms * 1e-3) / 1e9, peak=8000.0, unit="GB/s", frac=nbytes / (ms * 1e-3) / 1e9 / 8000.0)
    return out


def job_leg(states, POS, paint, g, L, N, device, args):
    """The whole job through the PRODUCT entry points (ldweaver_amd.mi, the mirror of the R functions), on its own engine:
    H2D of the state matrix, estimate_Hamming_distance_weights, perform_MI_computation (cold MI pass, lr_links.tsv, short-range
    model, ARACNE, sr_links.tsv).  What the reference's job contains (R/computePairwiseMI.R:46-145) and the bench's timed step
    leaves out: the upload and the two write.table calls (:140, :362)."""
    import shutil
    import tempfile

    import torch
    from ldweaver_amd import mi as MIH
    from ldweaver_amd.engine import Engine
    from ldweaver_amd.snpdat import CdsVar, SnpDat

    st_host = states.cpu().numpy()                       # pageable host memory, as R would hand it over
    tmp = tempfile.mkdtemp(prefix="ldw_job_")
    out = {}
    try:
        with Engine(device) as e2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e2.set_alignment(st_host)
            e2.sync()
            out["h2d_ms"] = (time.perf_counter() - t0) * 1e3
            out["h2d_GBps"] = st_host.nbytes / (out["h2d_ms"] * 1e-3) / 1e9
            counts = e2.state_counts()
            sd = SnpDat.from_states(st_host, POS, g, counts=counts)
            t1 = time.perf_counter()
            hdw = MIH.estimate_Hamming_distance_weights(sd, threshold=0.1, engine=e2, alignment_resident=True, verbose=False)
            out["hamming_weights_s"] = time.perf_counter() - t1
            t2 = time.perf_counter()
            red, aux = MIH.perform_MI_computation(sd, hdw, CdsVar(paint=paint, nclust=3), lr_save_path=os.path.join(tmp, "lr_links.tsv"),
                                                  sr_save_path=os.path.join(tmp, "sr_links.tsv"), plt_folder=os.path.join(tmp, "PLOTS"),
                                                  max_blk_sz=args.max_blk_sz, engine=e2, alignment_resident=True, verbose=False, return_aux=True)
            out["perform_MI_computation_s"] = time.perf_counter() - t2
            out["stages_s"] = aux.get("stages_s")
            out["tsv_write_s"] = aux["tsv_write_s"]
            out["lr_rows_written"] = aux["lr_rows_written"]
            out["sr_rows_written"] = int(len(red))
            out["lr_tsv_bytes"] = os.path.getsize(os.path.join(tmp, "lr_links.tsv")) if aux["lr_rows_written"] else 0
            out["sr_tsv_bytes"] = os.path.getsize(os.path.join(tmp, "sr_links.tsv"))
            out["path"] = aux["path"]
        out["job_s"] = out["h2d_ms"] * 1e-3 + out["hamming_weights_s"] + out["perform_MI_computation_s"]
        out["what"] = ("states on the host -> H2D -> estimate_Hamming_distance_weights -> perform_MI_computation (cold MI pass of all block pairs, "
                       "lr_links.tsv, short-range model, ARACNE, sr_links.tsv on disk); first use of this engine, so its allocations are inside")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def adversarial_leg(L, N, device, args):
    """what the default path costs on data that is NOT friendly to it.  Same L x N, MAF uniform in [0.2, 0.5] (no rare
    minor states: the marginal-only tile pruning has nothing to dismiss), no clonal groups (synth kind 'adversarial'), on an engine of
    its own, cold passes.  Two weightings: the one estimate_Hamming_distance_weights gives on that alignment (no sequence has a neighbour
    within 10 %: all weights equal, N_eff = N / 2) and N DISTINCT weights in [1/50, 1] set directly (the dual-digit GEMM then needs its
    per-32-position exponents).  Each with ms_per_step, path, prune, spec_misses, pairs listed, and the plain path beside it."""
    import torch
    from ldweaver_amd.engine import Engine
    from ldweaver_amd.mi import lr_links_approx, make_blocks
    from ldweaver_amd.synth import synth_alignment

    syn = synth_alignment(L, N, seed=1988, device=torch.device("cuda", device), as_numpy=False, kind="adversarial")
    POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
    blocks = make_blocks(L, args.max_blk_sz)
    pairs = 0
    for fs, fe, ts, te in blocks.tolist():
        nf, nt = fe - fs + 1, te - ts + 1
        pairs += nf * (nf - 1) // 2 if (fs == ts and fe == te) else nf * nt - min(nf, nt)
    approx = lr_links_approx(POS, g, 20000.0)
    out = dict(what="synthetic kind 'adversarial': MAF uniform in [0.2, 0.5], no clonal groups; cold passes on an engine of its own", L=L, N=N, pairs=int(pairs))
    with Engine(device) as e:
        e.set_alignment(syn["states"])
        counts = e.state_counts()
        uqe = (counts > 0).T.astype(np.float64)
        r = uqe.sum(axis=1)
        hdw_h = e.hamming_weights(int(L * 0.1))
        u = ((np.arange(N, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32)
        hdw_d = 1.0 / (1.0 + 49.0 * u)
        for tag, hdw in (("hamming_weights", hdw_h), ("distinct_weights", hdw_d)):
            e.set_weights(hdw)
            e.set_snp_meta(r, uqe, POS, paint, g)
            rec = dict(weights=dict(distinct=int(len(np.unique(hdw))), neff=float(hdw.sum()), min=float(hdw.min()), max=float(hdw.max())),
                       approximate_gemm=e.apx_info())

            def run(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    e.reset_speculation()
                    e.mi_all_pairs(blocks, 20000.0, 1e6, approx)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            for key, (mixed, scr, path) in (("default", (True, 1, 0)), ("plain", (False, 0, 1))):
                e.set_mixed(mixed)
                e.set_screen(scr)
                e.set_path(path)
                run(1)
                p0, c0, pr0, sp0, ov0 = e.path_report(), e.counters(), e.prune_report(), e.span_report(), e.overflow_report()
                ms = run(2 if key == "plain" else 3)
                p1, c1, pr1, sp1, ov1 = e.path_report(), e.counters(), e.prune_report(), e.span_report(), e.overflow_report()
                steps = 2 if key == "plain" else 3
                rec[key] = dict(ms_per_step=ms, value=pairs / (ms * 1e-3), steps=steps, links=dict(n_sr=e.links_count(0), n_lr=e.links_count(1)),
                                path={k: ((p1[k] - p0[k]) / steps if isinstance(p1[k], int) else p1[k]) for k in p1},
                                spec_misses=c1["spec_misses"] - c0["spec_misses"], screen_violations=c1["screen_violations"] - c0["screen_violations"],
                                prune=dict(tiles_pruned=pr1["tiles_pruned"] - pr0["tiles_pruned"], tiles_total=pr1["tiles_total"] - pr0["tiles_total"]),
                                spans=(sp1["spans"] - sp0["spans"]) / steps, span_blocks_redone=sp1["redone"] - sp0["redone"],
                                list_overflows=dict(pair_list=ov1["pair_list"] - ov0["pair_list"], maybe_list=ov1["maybe_list"] - ov0["maybe_list"]))
            e.set_mixed(True)
            e.set_screen(1)
            e.set_path(0)
            rec["links_equal_plain"] = rec["default"]["links"] == rec["plain"]["links"]
            # r05: the guard that was missing — the default path has to beat the plain path here too, without a
            # single block redone; anything else is said on the line in words
            d = rec["default"]
            rec["default_over_plain"] = d["ms_per_step"] / rec["plain"]["ms_per_step"]
            bad = []
            if rec["default_over_plain"] > 1.0:
                bad.append(f"REGRESSION: default path {d['ms_per_step']:.1f} ms is SLOWER than the plain path {rec['plain']['ms_per_step']:.1f} ms")
            if d["spec_misses"] or d["span_blocks_redone"] or d["list_overflows"]["pair_list"] or d["list_overflows"]["maybe_list"]:
                bad.append(f"REGRESSION: {d['spec_misses']} blocks redone ({d['span_blocks_redone']} span segments; list overflows {d['list_overflows']}) in {d['steps']} cold passes")
            if not rec["links_equal_plain"]:
                bad.append("WRONG: link counts differ from the plain path")
            rec["note"] = "; ".join(bad) if bad else "ok: default faster than plain, 0 blocks redone, 0 list overflows, links equal"
            out[tag] = rec
    out["ok"] = all(out[t]["note"].startswith("ok") for t in ("hamming_weights", "distinct_weights"))
    return out


def sr_tail_leg(eng, blocks, mine, sr_dist, lr_retain, approx, POS, paint, g, dev, rank, world, fence):
    """N > 1, after the timed region: pass + short-range model + ARACNE, (a) the r04 route — link tables gathered to rank 0
    (short-range rows as their MI column), rank 0 runs the model alone on the assembled 1.44 GB table — against (b) the rows LEFT on their ranks
    (dist_srp.merge_n_sort_sr_links_dist: bounds, ~7 % of the MI column, block sums, kept links and pool travel; only the long-range table is gathered).
    Same kept links, srp and flags; per-rank bytes on the line.  Both wall clocks include the pass itself."""
    import torch
    import torch.distributed as dist
    from ldweaver_amd.dist import gather_block_stats, gather_link_tables
    from ldweaver_amd.dist_srp import merge_n_sort_sr_links_dist
    from ldweaver_amd.srp import merge_n_sort_sr_links_device
    nblocks, nclust, cut = len(blocks), int(np.max(paint)), 3.0

    def run_pass():
        eng.reset_speculation()
        if len(mine):
            eng.mi_all_pairs(blocks[mine], sr_dist, lr_retain, approx)
            return eng.block_stats()
        eng.links_begin(1)
        eng.links_end()
        z = np.zeros(0, dtype=np.int64)
        return dict(n_sr=z, n_lr_kept=z, n_lr_total=z, disc_thresh=np.zeros(0))

    def digest(red, flags):
        return dict(rows=int(len(red["MI"])), aracne_true=int(np.sum(flags)), mi_sum=float(np.sum(red["MI"])), srp_sum=float(np.sum(red["srp_max"])),
                    first=[int(red["a"][0]), int(red["b"][0])] if len(red["MI"]) else None)

    res = {}
    # (a) gather, then the model on rank 0
    fence()
    t0 = time.perf_counter()
    st = run_pass()
    out = gather_link_tables({"sr": eng.links_view(0), "lr": eng.links_view(1)}, mine, {"sr": st["n_sr"], "lr": st["n_lr_kept"]}, nblocks,
                             sr_pairs=lambda n: eng.sr_pairs(blocks, sr_dist, n))
    sent_a = int(8 * np.sum(st["n_sr"]) + 16 * np.sum(st["n_lr_kept"])) if rank != 0 else 0
    fence()
    t1 = time.perf_counter()
    dig_a = None
    stats_all = gather_block_stats(st, mine, nblocks)
    if rank == 0:
        eng.links_import(0, *out["sr"])
        eng.links_import(1, *out["lr"])
        red, flags, _ = merge_n_sort_sr_links_device(eng, nclust, sr_dist, cut, POS, paint, g, run_aracne=True, order_links=True, block_rows=stats_all["n_sr"])
        dig_a = digest(red, flags)
    del out
    fence()
    t2 = time.perf_counter()
    res["gather"] = dict(pass_and_gather_ms=(t1 - t0) * 1e3, model_aracne_on_rank0_ms=(t2 - t1) * 1e3, total_ms=(t2 - t0) * 1e3)
    # (b) the rows stay
    fence()
    t0 = time.perf_counter()
    st = run_pass()
    stats_all = gather_block_stats(st, mine, nblocks)
    e = lambda dt: torch.empty(0, dtype=dt, device=dev)
    out = gather_link_tables({"sr": (e(torch.int32), e(torch.int32), e(torch.float64)), "lr": eng.links_view(1)}, mine,
                             {"sr": np.zeros(len(mine), dtype=np.int64), "lr": st["n_lr_kept"]}, nblocks)
    fence()
    t1 = time.perf_counter()
    red, flags, aux = merge_n_sort_sr_links_dist(eng, nclust, sr_dist, cut, POS, paint, g, mine, stats_all["n_sr"], run_aracne=True, order_links=True)
    fence()
    t2 = time.perf_counter()
    sent_b = dict(aux["bytes_sent"], lr_table=int(16 * np.sum(st["n_lr_kept"])) if rank != 0 else 0)
    res["dist"] = dict(pass_and_lr_gather_ms=(t1 - t0) * 1e3, model_aracne_over_ranks_ms=(t2 - t1) * 1e3, total_ms=(t2 - t0) * 1e3)
    mine_rec = dict(rank=rank, sr_rows=int(np.sum(st["n_sr"])), candidates=int(aux["candidates"]), bytes_sent_gather=sent_a, dist_ms_by_step={k: round(v, 2) for k, v in aux["ms"].items()},
                    bytes_sent_dist=int(sum(sent_b.values())), bytes_sent_dist_by_exchange=sent_b)
    recs = [None] * world
    dist.all_gather_object(recs, mine_rec)
    if rank != 0:
        return None
    dig_b = digest(red, flags)
    res["per_rank"] = recs
    res["kept_links"] = dict(gather=dig_a, dist=dig_b)
    same = dig_a == dig_b   # (bit for bit: both routes sum the excess statistics per reference block in make_blocks order)
    res["kept_links_equal"] = bool(same)
    peers = [r for r in recs if r["rank"] != 0]
    res["max_bytes_sent_per_peer"] = dict(gather=max(r["bytes_sent_gather"] for r in peers), dist=max(r["bytes_sent_dist"] for r in peers))
    res["what"] = ("pass + mergeNsort_sr_links + runARACNE behind it, wall clock incl. the pass: `gather` = link tables assembled on rank 0 (short-range rows as their MI "
                   "column), model on rank 0 alone; `dist` = short-range rows left on their ranks (only bounds, the rows from the smallest local 95 % order statistic up, "
                   "block sums, kept links and the ARACNE pool travel; the long-range table is gathered as before).  All ranks share ONE GPU when the backend is gloo: "
                   "times there say nothing about xGMI (unmeasured on hardware); the bytes do")
    return res


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), relay rank 0's JSON line and the
    child's exit code.  Runs before anything has touched torch or the GPU in this process (never re-exec a process that has: the
    child is a plain subprocess, this process only waits for it)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), LDW_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"# bench.py --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for l in p.stdout:
        if l.startswith("{") and '"metric"' in l:
            line = l.rstrip("\n")
        else:
            sys.stderr.write(l)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("# bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def inproc_main(args):
    """`--inproc`: the block loop over several contexts of THIS process (ldw_mi_all_pairs_multi) — one context per device,
    worker threads and the peer-to-peer gather inside the library, no torch.distributed.  Same workload, same cold steps, same line;
    `inproc` carries the slowest pass, the gather and every context's own pass time."""
    import torch
    from ldweaver_amd.engine import Engine
    from ldweaver_amd.mi import lr_links_approx, make_blocks
    from ldweaver_amd.synth import synth_alignment
    devs = [int(x) for x in args.inproc_devices.split(",")] if args.inproc_devices else list(range(args.gpus))
    L, N = args.L, args.N
    torch.cuda.set_device(devs[0])
    syn = synth_alignment(L, N, seed=1988, device=torch.device("cuda", devs[0]), as_numpy=False)
    POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
    blocks = make_blocks(L, args.max_blk_sz)
    pairs = 0
    for fs, fe, ts, te in blocks.tolist():
        nf, nt = fe - fs + 1, te - ts + 1
        pairs += nf * (nf - 1) // 2 if (fs == ts and fe == te) else nf * nt - min(nf, nt)
    approx = lr_links_approx(POS, g, 20000.0)
    engs = [Engine(d) for d in devs]
    try:
        st_host = syn["states"].cpu().numpy()
        for e in engs:
            e.set_alignment(syn["states"] if e.device == devs[0] else st_host)
        counts = engs[0].state_counts()
        uqe = (counts > 0).T.astype(np.float64)
        r = uqe.sum(axis=1)
        t0 = time.perf_counter()
        hdw = Engine.hamming_weights_multi(engs, int(L * 0.1))
        hamming_s = time.perf_counter() - t0
        for e in engs:
            e.set_weights(hdw, args.nlimbs)
            e.set_snp_meta(r, uqe, POS, paint, g)
        info = {}

        def step():
            for e in engs:
                e.reset_speculation()
            info.update(Engine.mi_all_pairs_multi(engs, blocks, 20000.0, 1e6, approx))

        def sync():
            for d in set(devs):
                torch.cuda.synchronize(d)
        for _ in range(args.warmup):
            step()
        sync()
        acc = dict(pass_ms=0.0, gather_ms=0.0, per_engine_ms=np.zeros(min(len(engs), 8)))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            acc["pass_ms"] += info["pass_ms"]
            acc["gather_ms"] += info["gather_ms"]
            acc["per_engine_ms"] += np.asarray(info["per_engine_ms"])
        sync()
        dt = time.perf_counter() - t0
        K = args.steps
        out = dict(metric="MI SNP-pairs/sec", value=pairs * K / dt, unit="pairs/s", n_gpus=len(set(devs)), steps=K, warmup=args.warmup, ms_per_step=dt / K * 1e3,
                   higher_is_better=True, scaling="strong", vs_baseline=None, dtype="i8", data="synthetic",
                   config=dict(workload=f"synthetic {L} SNPs x {N} seqs, all {len(blocks)} block pairs of make_blocks(max_blk_sz={args.max_blk_sz}), sr_dist=20000, "
                                        f"lr_retain_links=1e6, sr+lr link tables assembled in context 0", L=L, N=N, pairs=int(pairs),
                               step="cold pass per step (ldw_reset_speculation on every context)",
                               parallelism=f"ONE process, {len(engs)} contexts on devices {devs}: ldw_mi_all_pairs_multi (worker threads, peer-to-peer gather)"),
                   inproc=dict(contexts=len(engs), devices=devs, blocks_per_context=np.bincount(info["owner"], minlength=len(engs)).tolist(),
                               slowest_pass_ms=acc["pass_ms"] / K, gather_ms=acc["gather_ms"] / K, per_context_pass_ms=(acc["per_engine_ms"] / K).tolist(),
                               hamming_weights_s=hamming_s),
                   links=dict(n_sr=engs[0].links_count(0), n_lr=engs[0].links_count(1)),
                   spec_misses=sum(e.counters()["spec_misses"] for e in engs))
        # r05, after the timed region: pass + short-range model + ARACNE both ways — tables gathered into context 0 and the model there, or the short-range
        # rows left on their contexts (LDW_MI_SR_ROWS_STAY) and the model's reductions over the contexts (ldw_sr_*_multi)
        if len(engs) > 1 and not args.no_sr_tail_leg:
            from ldweaver_amd.engine import EngineGroup
            from ldweaver_amd.srp import merge_n_sort_sr_links_device
            nclust, leg = int(np.max(paint)), {}
            dig = {}
            for tag, stay in (("gather", False), ("rows_stay", True)):
                for e in engs:
                    e.reset_speculation()
                sync()
                t0 = time.perf_counter()
                inf = Engine.mi_all_pairs_multi(engs, blocks, 20000.0, 1e6, approx, sr_rows_stay=stay)
                t1 = time.perf_counter()
                view = EngineGroup(engs) if stay else engs[0]
                red, flags, _ = merge_n_sort_sr_links_device(view, nclust, 20000.0, 3.0, POS, paint, g, run_aracne=True, order_links=True,
                                                             block_rows=None if stay else engs[0].block_stats()["n_sr"])
                t2 = time.perf_counter()
                leg[tag] = dict(pass_and_gather_ms=(t1 - t0) * 1e3, gather_ms=inf["gather_ms"], model_aracne_ms=(t2 - t1) * 1e3, total_ms=(t2 - t0) * 1e3)
                dig[tag] = dict(rows=int(len(red["MI"])), aracne_true=int(np.sum(flags)), mi_sum=float(np.sum(red["MI"])), srp_sum=float(np.sum(red["srp_max"])))
            leg["kept_links"] = dig
            leg["kept_links_equal"] = dig["gather"] == dig["rows_stay"]
            out["sr_tail"] = leg
        print(json.dumps(out), flush=True)
    finally:
        for e in engs:
            e.close()
    return 0


def main():
    args = parse()
    if args.inproc:
        sys.exit(inproc_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist
    from ldweaver_amd import _lib as LL
    from ldweaver_amd.dist import deal_blocks, gather_begin, gather_end, gather_link_tables
    from ldweaver_amd.engine import Engine
    from ldweaver_amd.mi import lr_links_approx, make_blocks
    from ldweaver_amd.synth import synth_alignment

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":      # every rank shares GPU 0: exercises sharding + gather logic only
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    ranks_seen = dist.get_world_size() if world > 1 else 1          # the world as the communicator (RCCL for backend nccl) reports it
    backend_seen = dist.get_backend() if world > 1 else None

    # ---- synthetic workload (same seed on every rank -> identical replicated alignment) ----
    L, N = args.L, args.N
    t_setup = time.time()
    syn = synth_alignment(L, N, seed=1988, device=dev, as_numpy=False)
    states = syn["states"]
    POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
    stream = torch.cuda.Stream(device=dev)   # a real (non-null) stream shared by torch and the library
    torch.cuda.set_stream(stream)
    eng = Engine(local_rank, stream=stream.cuda_stream)
    eng.set_engine({"mfma": LL.ENGINE_MFMA, "hist": LL.ENGINE_HIST, "hist_states": LL.ENGINE_HIST_STATES}[args.engine])
    eng.set_overlap(not args.no_overlap)
    eng.set_prune(not args.no_prune)
    eng.set_fused(args.fused)
    eng.set_screen(args.screen)
    eng.set_mixed(not args.no_mixed)
    eng.set_path(args.path)
    eng.set_alignment(states)
    counts = eng.state_counts()
    uqe = (counts > 0).T.astype(np.float64)
    r = uqe.sum(axis=1)
    torch.cuda.synchronize()
    eng.gemm_stats(reset=True)
    t0 = time.time()
    hdw = eng.hamming_weights(int(L * 0.1))
    torch.cuda.synchronize()
    hamming_s = time.time() - t0
    hamming_kernel_ms = eng.last_timing()["gemm_ms"]
    roof_hamming = hamming_roofline(eng.hamming_stats(), eng.gemm_stats(reset=True), hamming_s, L, N)
    eng.set_weights(hdw, args.nlimbs)
    eng.set_snp_meta(r, uqe, POS, paint, g)
    sr_dist, lr_retain = 20000.0, 1e6
    approx = lr_links_approx(POS, g, sr_dist)
    blocks = make_blocks(L, args.max_blk_sz)
    nblocks = len(blocks)
    mine = deal_blocks(blocks, world)[rank]
    my_blocks = blocks[mine]
    setup_s = time.time() - t_setup

    # (the CPU baseline runs LAST, r04: between the set-up and the first pass its 20-30 s left the GPU idle, and the first pass after such a
    # pause took 74 ms against the 42 ms it takes when it follows the set-up at once, as it does in a job)
    cpu_base = None

    pairs = 0
    for fs, fe, ts, te in blocks.tolist():
        nf, nt = fe - fs + 1, te - ts + 1
        pairs += nf * (nf - 1) // 2 if (fs == ts and fe == te) else nf * nt - min(nf, nt)  # Q3: off-diagonal blocks drop their diagonal

    tim = dict(gemm_ms=0.0, epilogue_ms=0.0, select_ms=0.0, total_ms=0.0)
    result = {}

    def compute(sub, accumulate_timing):
        """This rank's blocks `sub` (indices into blocks) -> its link tables of those blocks (device tensors) + row counts."""
        if len(sub):
            eng.mi_all_pairs(blocks[sub], sr_dist, lr_retain, approx)
            st = eng.block_stats()
            # the tables as they lie in HBM (no copy): on one rank they ARE the assembled result; on several ranks gather_begin packs
            # them into its send buffer before the next blocks are computed
            local = {"sr": eng.links_view(0), "lr": eng.links_view(1)}
            cnt = {"sr": st["n_sr"], "lr": st["n_lr_kept"]}
            if accumulate_timing:
                for k, v in eng.last_timing().items():
                    tim[k] += v
        else:
            e = lambda dt: torch.empty(0, dtype=dt, device=dev)
            local = {k: (e(torch.int32), e(torch.int32), e(torch.float64)) for k in ("sr", "lr")}
            cnt = {"sr": np.zeros(0, dtype=np.int64), "lr": np.zeros(0, dtype=np.int64)}
        return local, cnt

    # N > 1: the gather runs in phases, so that the rows of the blocks a rank has finished travel over xGMI while it
    # computes its next ones; only the last phase's transfer is exposed.  Every rank runs the same number of phases.
    # (every phase is its own pass of the block pipeline, which costs a start-up / drain: at least --min-blocks-per-phase blocks per phase — phases_for)
    n_phase = phases_for(nblocks, world, args.gather_phases, args.min_blocks_per_phase)
    my_phases = np.array_split(mine, n_phase)

    rk_acc = dict(compute_ms=0.0, gather_begin_ms=0.0, exposed_gather_ms=0.0, bytes_sent=0, steps=0)
    cold = not args.warm

    def step(accumulate_timing, reset=None):
        if cold if reset is None else reset:
            eng.reset_speculation()   # a job's first (and only) pass: nothing inherited from an earlier pass over the same data
        if world == 1:
            local, cnt = compute(mine, accumulate_timing)
            out = gather_link_tables(local, mine, cnt, nblocks)   # (one rank: an in-place view, nothing moves)
        else:
            started = []
            for sub in my_phases:
                t0 = time.perf_counter()
                local, cnt = compute(sub, accumulate_timing)
                t1 = time.perf_counter()
                started.append(gather_begin(local, sub, cnt, nblocks, sr_index=not args.sr_mi_only))
                t2 = time.perf_counter()
                if accumulate_timing:
                    rk_acc["compute_ms"] += (t1 - t0) * 1e3
                    rk_acc["gather_begin_ms"] += (t2 - t1) * 1e3
                    if rank != 0:
                        rk_acc["bytes_sent"] += int(started[-1].mine.numel())
            t3 = time.perf_counter()
            # (--sr-mi-only, default: the short-range rows travel as their MI column alone; rank 0 rebuilds their index columns from the positions)
            out = gather_end(started, nblocks, sr_pairs=(lambda n: eng.sr_pairs(blocks, sr_dist, n)) if args.sr_mi_only else None)
            if accumulate_timing:
                rk_acc["exposed_gather_ms"] += (time.perf_counter() - t3) * 1e3   # waiting for transfers + (rank 0) re-interleaving the segments
                rk_acc["steps"] += 1
        if out is not None:
            result["n_sr"] = int(out["sr"][2].numel())
            result["n_lr"] = int(out["lr"][2].numel())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, accumulate=False, reset=None):
        """n steps bracketed by fences; wall seconds."""
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            step(accumulate, reset)
        fence()
        return time.perf_counter() - t0

    first_pass_ms = None
    for w in range(args.warmup):
        t = timed(1)
        if w == 0:
            first_pass_ms = t * 1e3     # includes the one-off allocations of the context and has no bucket guesses
    dt = timed(args.steps, accumulate=True)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.backend == "gloo" else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    counters_timed = eng.counters()
    links_timed = dict(result)
    per_rank = None
    if world > 1:
        mine_rec = dict(rank=rank, blocks=int(len(mine)), phases=int(n_phase),
                        **{k: (v / max(1, rk_acc["steps"]) if k != "steps" else v) for k, v in rk_acc.items()})
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)

    run_sr_tail = world > 1 and not args.no_extra_legs and not args.no_sr_tail_leg and args.engine == "mfma"   # (after the line is assembled: see the end of main)

    extra = rank == 0 and world == 1 and len(my_blocks) and not args.no_extra_legs
    legs = {}
    # ---- warm replay: K steps that inherit their predecessor's bucket guesses, spread history and threshold table (what r02
    # reported as its headline; a real job never sees this state) ----
    if extra and cold:
        timed(1, reset=False)
        t_w = timed(args.steps, reset=False)
        legs["warm_replay"] = dict(ms_per_step=t_w / args.steps * 1e3, value=pairs * args.steps / t_w, steps=args.steps, links=dict(result),
                                   note="steps WITHOUT ldw_reset_speculation: each inherits the previous identical step's guesses (not a job)")
    # ---- sustained: the same (cold) step until >= 10 s of steps have run (clock / thermal steady state; also what a coarse busy
    # sampler can see) ----
    if extra and args.sustain_s > 0:
        done_s, n_s = dt, args.steps
        t_chunk = 0.0
        n_chunk = 0
        while done_s + t_chunk < args.sustain_s:
            k = max(1, min(20, int((args.sustain_s - done_s - t_chunk) / max(1e-3, dt / args.steps)) + 1))
            t_chunk += timed(k)
            n_chunk += k
        if n_chunk:
            legs["sustained"] = dict(steps=n_chunk, seconds=t_chunk, ms_per_step=t_chunk / n_chunk * 1e3, value=pairs * n_chunk / t_chunk,
                                     note="further steps of the same kind right after the timed region")

    # ---- kernel-exclusive stage times for the roofline: in the timed region the GEMM of block b+1 runs on a second stream
    # beside the epilogue / selection of block b, so HIP-event brackets of one kernel also contain its neighbours.  Replay
    # the same step with the overlap off (same kernels, same inputs, back to back on one stream) and bracket there. ----
    tim_overlapped = dict(tim)
    n_replay, gst, serial_ms_per_step, cnt_replay, unpruned = 0, None, None, None, None
    if rank == 0 and len(my_blocks):
        eng.set_overlap(False)
        for k in tim:
            tim[k] = 0.0
        n_replay = min(args.steps, 3)
        eng.gemm_stats(reset=True)
        c0 = eng.counters()
        t_r0 = time.perf_counter()
        for _ in range(n_replay):
            eng.mi_all_pairs(my_blocks, sr_dist, lr_retain, approx)
            for k, v in eng.last_timing().items():
                tim[k] += v
        torch.cuda.synchronize()
        serial_ms_per_step = (time.perf_counter() - t_r0) / n_replay * 1e3
        gst = eng.gemm_stats(reset=True)
        c1 = eng.counters()
        cnt_replay = {k: c1[k] - c0[k] for k in c1}
        # the same kernel on FULL launches (tile pruning off): what its K loop delivers when every wave tile is computed
        unpruned = None
        if not args.no_prune and args.engine == "mfma" and extra:   # (not under --no-extra-legs: the profiled commands, whose per-kernel averages must be those of `roofline`)
            eng.set_prune(False)
            eng.mi_all_pairs(my_blocks, sr_dist, lr_retain, approx)
            eng.gemm_stats(reset=True)
            g_ms = 0.0
            for _ in range(n_replay):
                eng.mi_all_pairs(my_blocks, sr_dist, lr_retain, approx)
                g_ms += eng.last_timing()["gemm_ms"]
            torch.cuda.synchronize()
            g2 = eng.gemm_stats(reset=True)
            if g2["apx_launches"] > 0 and g_ms > 0:
                unpruned = dict(avg_launch_ms=g_ms / g2["apx_launches"], achieved=g2["apx_ops"] / (g_ms * 1e-3) / 1e12)
                unpruned["frac"] = unpruned["achieved"] / 5000.0
            eng.set_prune(True)
        eng.set_overlap(not args.no_overlap)

    # ---- mi_values_produced: no screen, no mixed precision, no approximate GEMM: 5-limb GEMM + fp64 MI of every pair ----
    if extra and args.engine == "mfma":
        eng.set_mixed(False)
        eng.set_screen(0)
        eng.set_path(1)
        timed(1)
        n_pl = max(10, min(args.steps, 20))   # (at least 10 steps, like the headline)
        t_pl = timed(n_pl)
        legs["mi_values_produced"] = dict(ms_per_step=t_pl / n_pl * 1e3, value=pairs * n_pl / t_pl, steps=n_pl, links=dict(result),
                                          what="--no-mixed --screen 0 --path 1: gemm_bits_kernel<5> + k_mi_epilogue_fast / _rest: an fp64 MI VALUE for every pair "
                                               "(the headline counts pairs DECIDED: the default path bounds 99.7 % of the pairs below their block's "
                                               "threshold instead of evaluating them; same link tables)")
        # its own roofline: the 5-limb exact GEMM, timed kernel-exclusively (overlap off) with HIP events, executed int8 operations counted by the library
        eng.set_overlap(False)
        eng.gemm_stats(reset=True)
        n_rp = 2
        g_ms = e_ms = s_ms = 0.0
        for _ in range(n_rp):
            eng.mi_all_pairs(my_blocks, sr_dist, lr_retain, approx)
            lt = eng.last_timing()
            g_ms, e_ms, s_ms = g_ms + lt["gemm_ms"], e_ms + lt["epilogue_ms"], s_ms + lt["select_ms"]
        torch.cuda.synchronize()
        g5 = eng.gemm_stats(reset=True)
        eng.set_overlap(not args.no_overlap)
        if g5["bits_launches"] > 0 and g_ms > 0:
            avg = g_ms / g5["bits_launches"]
            ach = g5["bits_ops"] / g5["bits_launches"] / (avg * 1e-3) / 1e12
            legs["roofline_mi_produced"] = dict(
                bound="mfma", kernel=f"gemm_bits_kernel<{args.nlimbs or 5}>", peak=5000.0, unit="TFLOP/s", achieved=ach, frac=ach / 5000.0,
                avg_launch_ms=avg, launches=g5["bits_launches"], executed_ops_per_launch=g5["bits_ops"] / g5["bits_launches"],
                alg_ops_per_launch=50.0 * N * pairs * n_rp / g5["bits_launches"],
                alg_frac=50.0 * N * pairs * n_rp / g5["bits_launches"] / (avg * 1e-3) / 1e12 / 5000.0,
                stages_ms_per_step=dict(gemm_ms=g_ms / n_rp, epilogue_ms=e_ms / n_rp, select_ms=s_ms / n_rp),
                epilogue=epilogue_roofline(e_ms / n_rp, pairs, (L, N, world)),
                hbm_alg_GBps=(L * N + 8.0 * pairs) / (t_pl / n_pl) / 1e9, hbm_frac=(L * N + 8.0 * pairs) / (t_pl / n_pl) / 1e9 / 8000.0,
                traffic=None,
                measured_in=f"{n_rp} serialized replay steps of the plain path (overlap off), HIP events around the kernels",
                note="achieved = executed int8 operations (2 x rows x rows x positions x 5 limbs of every workgroup tile that runs; one indicator row per minor "
                     "state: 1.16 rows per SNP) / launch time; alg_frac prices SURVEY 8(d)'s 50 N flops per pair against the same peak — above `frac` "
                     "because the row reduction removes 18x of the one-hot formulation's work and the 5 limbs put 5x back.  profiles/r06_c4_plain_serial_kernel_stats.csv "
                     "holds rocprofv3's average for the same kernel")
        tp = os.path.join(ROOT, "profiles", "r06_pmc_traffic_plain.json")   # (PMC passes of `bench.py --no-mixed --screen 0 --path 1`, tools/r06_final.sh)
        if "roofline_mi_produced" in legs and os.path.exists(tp) and (L, N, world) == (100_000, 5_000, 1):
            ent = json.load(open(tp)).get(legs["roofline_mi_produced"]["kernel"])
            if ent:
                legs["roofline_mi_produced"].update(traffic=ent.get("hbm_bytes_per_launch"), traffic_range=ent.get("hbm_bytes_per_launch_range"),
                                                    traffic_source="profiles/r06_pmc_traffic_plain.json (FETCH_SIZE x 2 + WRITE_SIZE per launch, separate --pmc passes)")
        eng.set_mixed(not args.no_mixed)
        eng.set_screen(args.screen)
        eng.set_path(args.path)
        legs["first_pass_incl_allocations_ms"] = first_pass_ms
    path_report = eng.path_report() if rank == 0 else None

    # ---- job: the product entry points end to end, state matrix on the HOST -> both tsv files on disk ----
    if extra and not args.no_job and args.engine == "mfma":
        legs["job"] = job_leg(states, POS, paint, g, L, N, local_rank, args)

    if extra and not args.no_adversarial and args.engine == "mfma":
        legs["adversarial"] = adversarial_leg(L, N, local_rank, args)

    # ---- a predicted strong-scaling curve on every line (rank 0 runs every share of the N = 1, 2, 4, 8 deals alone; the other ranks wait at the next fence) ----
    if rank == 0 and not args.no_extra_legs and args.engine == "mfma" and nblocks >= 8:
        legs["scaling_model"] = scaling_model_leg(eng, blocks, sr_dist, lr_retain, approx, pairs, args.gather_phases, world, dt / args.steps * 1e3, args.min_blocks_per_phase)

    if rank == 0:
        K = args.steps
        ms_per_step = dt / K * 1e3
        value = pairs * K / dt
        # ---- roofline of the dominant kernel: the block-wide co-occurrence GEMM (live HIP-event times of the replay) ----
        my_pairs = 0
        for fs, fe, ts, te in my_blocks.tolist():
            nf, nt = fe - fs + 1, te - ts + 1
            my_pairs += nf * (nf - 1) // 2 if (fs == ts and fe == te) else nf * nt - min(nf, nt)
        nb_mine = max(1, len(my_blocks))
        J = args.nlimbs or 5
        i8_peak = 5000.0  # TOP/s dense (MI355X_MICROARCH.md: i8 = 2 x bf16 per clock, bf16 ~2.5 PF dense)
        stage = max(("gemm_ms", "epilogue_ms", "select_ms"), key=lambda k: tim[k])
        roof = dict(bound="mfma", peak=i8_peak, unit="TFLOP/s", traffic=None)
        if gst and (gst["apx_launches"] + gst["bits_launches"]) > 0 and tim["gemm_ms"] > 0:
            n_launch = gst["apx_launches"] + gst["bits_launches"]
            apx = gst["apx_launches"] >= gst["bits_launches"]
            mixed_blocks = cnt_replay.get("mixed_blocks", 0)
            fine = "per 32" in (path_report or {}).get("apx_gate", "")
            apx_name = "gemm_apx_lds_kernel" if os.environ.get("LDW_APX_KERNEL", "r")[:1] == "l" else f"gemm_apx_kernel<4, 2, {'true' if fine else 'false'}, 2>"   # (MT, NT, FINE, waves per SIMD: the name rocprofv3 reports)
            kname = apx_name if apx else (f"gemm_mi_fused_kernel<{J}>" if args.fused else f"gemm_bits_kernel<{3 if mixed_blocks else J}>")
            avg_ms = tim["gemm_ms"] / n_launch
            exec_per_launch = (gst["apx_ops"] + gst["bits_ops"]) / n_launch
            alg_per_launch = 50.0 * N * my_pairs * n_replay / n_launch      # 50 * N MAC-flops per pair
            achieved = exec_per_launch / (avg_ms * 1e-3) / 1e12
            pr = eng.prune_report()
            prune_note = ""
            if pr.get("on") and pr.get("tiles_total", 0) > 0:   # every figure of the note is this run's own (no numbers of another shape)
                prune_note = (f"Tile pruning (`prune`): since this engine was created the kernel ran on {pr['ordered_blocks']} weight-ordered launches over the "
                              f"list of the wave tiles whose pairs are NOT all dismissed by their marginals alone (k_apx_live_tiles): "
                              f"{pr['tiles_pruned']} of {pr['tiles_total']} wave tiles = {100.0 * pr['tiles_pruned'] / pr['tiles_total']:.1f} % pruned; pruned tiles are "
                              f"not in `achieved`, and the shorter launches pay a larger share of ramp and tail" +
                              (f": the same K loop on full launches (`full_launches_without_pruning`: the same replay with ldw_set_prune(0), measured "
                               f"live) runs at {unpruned['frac']:.2f} of the peak (docs/HISTORY.md 5.1c/d).  " if unpruned else ".  "))
            roof.update(kernel=kname, achieved=achieved, frac=achieved / i8_peak, avg_launch_ms=avg_ms, launches=n_launch,
                        executed_ops_per_launch=exec_per_launch,
                        alg_work_reduction=alg_per_launch / exec_per_launch,
                        alg_TFLOPs=alg_per_launch / (avg_ms * 1e-3) / 1e12,
                        launch_mix=dict(gemm_apx=gst["apx_launches"], gemm_bits_full=gst["bits_launches"], gemm_bits_band=gst["band_launches"]),
                        full_launches_without_pruning=unpruned,
                        overlapped_avg_launch_ms=tim_overlapped["gemm_ms"] / max(1, len(my_blocks) * K),
                        measured_in=f"{n_replay} serialized replay step(s) after the timed region (overlap off; the replay inherits the bucket guesses of the "
                                    f"pass before it, so that every block launches this kernel), {serial_ms_per_step:.2f} ms/step",
                        note="achieved = int8 operations the kernel EXECUTES per launch (2 x rows x rows x positions of every wave tile "
                             "that runs, counted by the library: ldw_gemm_stats) / its average launch time (HIP events around the kernel, "
                             "overlap off) ; frac = achieved / 5 POP/s dense int8.  alg_work_reduction = SURVEY 8(d)'s algorithmic 50*N "
                             "flops per pair x pairs per launch / executed operations: one indicator row per minor state instead of 5 one-hot "
                             "planes, " + ("one dual-digit pass instead of 5 limbs (the exact sums of what the screen lists come from "
                                           "k_pair_sums / the band GEMM)" if apx else f"{3 if mixed_blocks else J} int8 limbs") +
                             ".  The launch time of gemm_apx_kernel includes its epilogue, which applies the screen's threshold table to the "
                             "kernel's own accumulators (the regions that pass are neither stored nor screened).  " + prune_note +
                             "`overlapped_avg_launch_ms` "
                             "is the bracket inside the timed region, where the GEMM shares the GPU with the previous block's screens and selection")
            tpath = os.path.join(ROOT, "profiles", "r06u_pmc_traffic.json")   # rocprofv3 cannot run inside the bench: quoted only for
            if not os.path.exists(tpath):                                    # (r06u: the code as committed; r06t: before the deferred pair-list appends)
                tpath = os.path.join(ROOT, "profiles", "r06t_pmc_traffic.json")
            if os.path.exists(tpath) and (L, N, world) == (100_000, 5_000, 1):   # the configuration the PMC passes were collected on
                ent = json.load(open(tpath)).get(kname)
                if ent:
                    roof["traffic"] = ent.get("hbm_bytes_per_launch")
                    roof["traffic_range"] = ent.get("hbm_bytes_per_launch_range")
                    roof["traffic_source"] = (os.path.relpath(tpath, ROOT) + " (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, per launch; `traffic` "
                                              "takes FETCH_SIZE x 2 as the guide prescribes for gfx950, `traffic_range` = [raw, x 2])")
        else:
            roof.update(kernel={"hist": "k_cooc_popc", "hist_states": "k_mi_hist"}.get(args.engine), achieved=None, frac=None)
        roof["dominant_stage"] = stage
        roof["hbm_alg_GBps_whole_step"] = (L * N + 8.0 * pairs) / (dt / K) / 1e9
        roof["hbm_frac_whole_step"] = roof["hbm_alg_GBps_whole_step"] / 8000.0
        mixed = (J == 5) and not args.no_mixed and not args.fused and args.screen > 0
        out = dict(metric="MI SNP-pairs/sec", value=value, unit="pairs/s", n_gpus=world, steps=K, warmup=args.warmup,
                   ms_per_step=ms_per_step, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="i8",
                   data="synthetic",
                   config=dict(workload=f"synthetic {L} SNPs x {N} seqs, all {nblocks} block pairs of make_blocks(max_blk_sz={args.max_blk_sz}), "
                                        f"sr_dist=20000, lr_retain_links=1e6, sr+lr link tables on rank 0",
                               L=L, N=N, pairs=int(pairs), engine=args.engine, nlimbs=args.nlimbs or 5,
                               arithmetic="screen: one dual-digit int8 MFMA pass (rigorous error bound) -> fp32 bound; every emitted MI: exact "
                                          "int64 fixed-point joint sums -> f64",
                               value_counts="pairs DECIDED: every pair's MI is either produced (short-range pairs, long-range candidates: exact fp64) or "
                                            "rigorously bounded below its block's long-range threshold; link tables identical to producing every MI "
                                            "(`value_mi_produced`: the rate at which an fp64 MI of EVERY pair is produced, SURVEY 8(d)'s P)",
                               step=("cold pass: speculation state reset before every step (ldw_reset_speculation), buffers allocated" if cold else
                                     "WARM replay (--warm): every step inherits the previous step's bucket guesses"),
                               result_at_rank0=("in-place view of the engine's device-resident link tables (N = 1: no gather, no copy)" if world == 1 else
                                                ("tables assembled on rank 0 by the phased gather" +
                                                 (": short-range rows travel as their MI column (8 B), rank 0 rebuilds their index columns from the positions" if args.sr_mi_only else ""))),
                               fused=bool(args.fused), screen=args.screen, mixed_precision=bool(mixed), path=args.path,
                               approximate_gemm=eng.apx_info(),
                               parallelism=f"pair-space blocks over {world} GPU(s)"),
                   roofline=roof)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or (min(10_000, L // 2) if args.cpu_baseline_full else max(200, min(2000, L // 2)))   # default: ~10-30 s of host work at N = 5000
            st_np = states[: 2 * sample].cpu().numpy()
            cpu_base = cpu_baseline(st_np, hdw, r[: 2 * sample], uqe[: 2 * sample], N, sample)
            del st_np
        if cpu_base is not None:
            # the STATED sample (one diagonal + one off-diagonal 10 000 x 10 000 block, 235 s on these cores) is too long for the default line; the
            # committed run of it is quoted beside the bounded sample so that the two can be compared on the line itself (`--cpu-baseline-full` re-runs it)
            ref = os.path.join(ROOT, "profiles", "r05_c4_bench_cpu_full.json")
            if not cpu_base.get("sample_is_stated_block") and os.path.exists(ref) and (L, N) == (100_000, 5_000):
                try:
                    full = json.loads([l for l in open(ref) if l.startswith("{")][-1])["cpu_baseline"]
                    cpu_base["stated_sample_committed_run"] = dict(value=full["value"], cores=full["cores"], sample_s=full["sample_s"], sample_pairs=full["sample_pairs"],
                                                                   source="profiles/r05_c4_bench_cpu_full.json (bench.py --cpu-baseline-full)",
                                                                   ratio_bounded_over_stated=cpu_base["value"] / full["value"])
                except (OSError, ValueError, KeyError, IndexError):
                    pass
            out["cpu_baseline"] = cpu_base
        out.update(legs)
        if "mi_values_produced" in legs:   # SURVEY 8(d)'s metric — pairs whose MI is PRODUCED — as a first-class number beside the job rate
            out["value_mi_produced"] = legs["mi_values_produced"]["value"]
            out["ms_per_step_mi_produced"] = legs["mi_values_produced"]["ms_per_step"]
        out["spec_misses"] = counters_timed["spec_misses"]
        out["path"] = path_report
        out["prune"] = dict(eng.prune_report(), what="wave tiles of the approximate GEMM whose pairs the threshold table dismisses whatever their joint "
                                                     "count (rows ordered by minor-state weight): flagged clean without being computed; not in roofline.achieved")
        out["ranks_seen"] = ranks_seen
        out["backend"] = backend_seen
        out["self_launched"] = bool(os.environ.get("LDW_BENCH_SELF_LAUNCHED"))
        if per_rank is not None:
            out["per_rank"] = per_rank
        out.update(stages_ms_per_step={k: v / max(1, n_replay) for k, v in tim.items()},
                   stages_ms_per_step_overlapped={k: v / K for k, v in tim_overlapped.items()},
                   links=links_timed, counters=counters_timed, counters_replay=cnt_replay, hamming_weights_s=hamming_s,
                   hamming_gemm_ms=hamming_kernel_ms, roofline_hamming=roof_hamming, setup_s=setup_s)
    # ---- N > 1, LAST: the short-range model behind the pass, table gathered / rows left on their ranks (sr_tail_leg).  Everything the line holds is
    # assembled by now, and a watchdog on every rank bounds the leg: its exchanges have only run under gloo and on ONE GPU under RCCL (tests/rccl_worker.py), and
    # a rank stuck in a collective must not cost the line of the timed region — after --sr-tail-timeout seconds rank 0 prints the line without the leg and
    # every rank leaves (os._exit: a thread blocked inside a collective cannot be interrupted).
    if run_sr_tail:
        import threading

        line_lock = threading.Lock()
        line_state = dict(printed=False)

        def print_line_once():
            # (the timer can fire between the leg returning and dog.cancel(): one line, whoever gets here first)
            with line_lock:
                if not line_state["printed"] and rank == 0:
                    print(json.dumps(out), flush=True)
                line_state["printed"] = True

        def bail():
            with line_lock:
                if line_state["printed"]:
                    return
            if rank == 0:
                out["sr_tail"] = dict(note=f"the leg did not finish within {args.sr_tail_timeout} s on this line and was abandoned (unmeasured on hardware)")
            print_line_once()
            os._exit(3 if args.strict_exit else 0)   # the line of the timed region is out and says the leg was abandoned; --strict-exit: the launcher sees a failure too

        dog = threading.Timer(args.sr_tail_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            res = sr_tail_leg(eng, blocks, mine, sr_dist, lr_retain, approx, POS, paint, g, dev, rank, world, fence)
            dog.cancel()
            if rank == 0:
                out["sr_tail"] = res
        except BaseException as e:   # noqa: BLE001 — the line matters more than the leg; the other ranks leave through their own watchdogs
            dog.cancel()
            if rank == 0:
                # (a peer whose watchdog fires first closes its connections: rank 0 then sees an error of the transport before its own timer)
                out["sr_tail"] = dict(note=f"the leg was abandoned by a peer's watchdog ({args.sr_tail_timeout} s) or failed on rank 0 ({type(e).__name__}: {e}); unmeasured on hardware")
            print_line_once()
            os._exit(3 if args.strict_exit else 0)
        print_line_once()
    elif world > 1 and rank == 0:
        out["sr_tail"] = dict(note="not run on this line (--no-extra-legs / --no-sr-tail-leg)")
    if rank == 0 and not run_sr_tail:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
