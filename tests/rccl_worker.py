"""Worker of tests/test_gpu_parity.py::test_rccl_single_rank_walk: ONE rank under backend "nccl" (= RCCL) with the collectives
forced (LDW_FORCE_COLLECTIVE=1), so that everything ldweaver_amd/dist.py does under N > 1 — device-tensor all-reduces, the
packed byte buffers, the grouped isend / irecv (here a loop-back rank 0 -> rank 0), the phased gather, the sharded Hamming
strips, the error agreement and perform_MI_computation on top of them — executes on GPU tensors on one GPU.

    rccl_worker.py OUTDIR
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldweaver_amd import dist as D  # noqa: E402
from ldweaver_amd import mi as MIH  # noqa: E402
from ldweaver_amd.engine import Engine  # noqa: E402
from ldweaver_amd.snpdat import CdsVar, SnpDat  # noqa: E402


def fake_links(bi, kind, dev):
    rng = np.random.default_rng(1000 * bi + (7 if kind == "lr" else 3))
    n = int(rng.integers(0, 5000)) if bi % 5 else 0
    return (torch.as_tensor(rng.integers(0, 10 ** 6, n).astype(np.int32), device=dev),
            torch.as_tensor(rng.integers(0, 10 ** 6, n).astype(np.int32), device=dev), torch.as_tensor(rng.random(n), device=dev))


def main():
    outdir = sys.argv[1]
    os.environ["LDW_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    report = {}
    try:
        # -- dist.py function by function, GPU tensors ---------------------------------------------------------------------
        D.agree(True, None, "walk")
        blocks = MIH.make_blocks(23000, 4000)
        mine = D.deal_blocks(blocks, 1)[0]
        assert mine.tolist() == list(range(len(blocks)))

        def tables(sub):
            lo, cn = {}, {}
            for kind in ("sr", "lr"):
                segs = [fake_links(int(bi), kind, dev) for bi in sub]
                cn[kind] = np.array([len(s[2]) for s in segs], dtype=np.int64)
                e = lambda dt: torch.empty(0, dtype=dt, device=dev)
                lo[kind] = tuple(torch.cat([s[j] for s in segs]) if segs else e(dt)
                                 for j, dt in enumerate((torch.int32, torch.int32, torch.float64)))
            return lo, cn

        whole, cn = tables(mine)
        out = D.gather_link_tables(whole, mine, cn, len(blocks))          # all-reduce + loop-back isend/irecv
        for kind in ("sr", "lr"):
            for j in range(3):
                assert out[kind][j].is_cuda and torch.equal(out[kind][j], whole[kind][j]), (kind, j)
        phases = []
        for p0, p1 in ((0, 2), (2, 2), (2, len(mine))):                   # three phases, one of them empty
            lo, c = tables(mine[p0:p1])
            phases.append(D.gather_begin(lo, mine[p0:p1], c, len(blocks)))
        out3 = D.gather_end(phases, len(blocks))
        for kind in ("sr", "lr"):
            for j in range(3):
                assert torch.equal(out3[kind][j], whole[kind][j]), (kind, j)
        report["gather_rows"] = {k: int(len(out[k][2])) for k in out}
        st = dict(n_lr_total=np.arange(len(blocks)) + 100, n_lr_kept=np.arange(len(blocks)) + 10, n_sr=np.arange(len(blocks)) * 3,
                  disc_thresh=np.where(np.arange(len(blocks)) % 4 == 0, np.nan, 0.25 + np.arange(len(blocks))))
        allst = D.gather_block_stats(st, mine, len(blocks))
        assert allst["n_sr"].tolist() == st["n_sr"].tolist() and np.allclose(allst["disc_thresh"], np.nan_to_num(st["disc_thresh"]))
        try:
            D.agree(False, None, "a failing share")
            raise AssertionError("agree(False) did not raise")
        except RuntimeError as e:
            assert "rank(s) [0]" in str(e)

        # -- the reference-facing functions on top of it --------------------------------------------------------------------
        g = np.load(os.path.join(ROOT, "tests", "golden", "snp_sample_states.npz"))
        o = np.load(os.path.join(ROOT, "tests", "golden", "snp_sample_oracle.npz"))
        sd = SnpDat.from_states(g["states"], g["POS"], float(o["g"]))
        with Engine(0) as eng:
            hdw = MIH.estimate_Hamming_distance_weights(sd, threshold=0.1, engine=eng)      # strips + all-reduce on the device
            assert np.array_equal(hdw, o["hdw"])
            for sr_only in (False, True):
                tag = "sr" if sr_only else "full"
                red = MIH.perform_MI_computation(sd, o["hdw"], CdsVar(paint=o["paint"], nclust=3), ncores=1,
                                                 lr_save_path=os.path.join(outdir, f"lr_{tag}.tsv"), sr_save_path=os.path.join(outdir, f"sr_{tag}.tsv"),
                                                 plt_folder=os.path.join(outdir, "PLOTS"), max_blk_sz=1000, lr_retain_links=1e5, engine=eng,
                                                 perform_SR_analysis_only=sr_only, sr_dist=(3000 if sr_only else 20000), verbose=False, quirk_mode=1)
                red.to_pickle(os.path.join(outdir, f"red_{tag}.pkl"))
                report[f"n_red_{tag}"] = int(len(red))
            # r05: the short-range model over ranks (dist_srp.py) through the same backend — all-reduces, broadcast and the three variable-length
            # gathers on GPU tensors; only the long-range table is gathered
            red, aux = MIH.perform_MI_computation(sd, o["hdw"], CdsVar(paint=o["paint"], nclust=3), ncores=1,
                                                  lr_save_path=os.path.join(outdir, "lr_rows_stay.tsv"), sr_save_path=os.path.join(outdir, "sr_rows_stay.tsv"),
                                                  plt_folder=os.path.join(outdir, "PLOTS_rows_stay"), max_blk_sz=1000, lr_retain_links=1e5, engine=eng,
                                                  verbose=False, quirk_mode=1, sr_tail="dist", return_aux=True)
            red.to_pickle(os.path.join(outdir, "red_rows_stay.pkl"))
            report["sr_tail_bytes_sent"] = aux["stages_s"]["sr_tail_bytes_sent"]
        report["backend"] = dist.get_backend()
        json.dump(report, open(os.path.join(outdir, "report.json"), "w"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
