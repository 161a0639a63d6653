"""Worker of tests/test_gpu_parity.py::test_perform_mi_computation_two_ranks (and its SR-only / failing-rank variants): one
rank of a torch.distributed (gloo) run of perform_MI_computation, every rank with its own engine (on the one GPU of the test
box).  Rank 0 writes the files.

    dist_worker.py OUTDIR [full | sr_only | fail | rows_stay | sr_only_rows_stay]      (rows_stay: sr_tail="dist", the short-range rows are not gathered)
"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldweaver_amd.engine import Engine  # noqa: E402
from ldweaver_amd import mi as MIH  # noqa: E402
from ldweaver_amd.snpdat import CdsVar, SnpDat  # noqa: E402


def main():
    outdir = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "full"
    dist.init_process_group("gloo")
    g = np.load(os.path.join(ROOT, "tests", "golden", "snp_sample_states.npz"))
    o = np.load(os.path.join(ROOT, "tests", "golden", "snp_sample_oracle.npz"))
    sd = SnpDat.from_states(g["states"], g["POS"], float(o["g"]))
    if mode == "fail":
        # rank 1 cannot compute (no weights of the right length): BOTH ranks must raise instead of rank 0 waiting in the gather
        with Engine(0) as eng:
            eng.set_alignment(sd.states)
            bad = o["hdw"][:-3] if dist.get_rank() == 1 else o["hdw"]
            try:
                MIH.perform_MI_computation(sd, bad, CdsVar(paint=o["paint"], nclust=3), lr_save_path=os.path.join(outdir, "lr.tsv"),
                                           sr_save_path=os.path.join(outdir, "sr.tsv"), plt_folder=os.path.join(outdir, "P"), max_blk_sz=1000,
                                           engine=eng, alignment_resident=True, verbose=False)
            except Exception as e:
                open(os.path.join(outdir, f"raised_{dist.get_rank()}.txt"), "w").write(f"{type(e).__name__}: {e}")
        dist.barrier()
        dist.destroy_process_group()
        return
    with Engine(0) as eng:
        hdw = MIH.estimate_Hamming_distance_weights(sd, threshold=0.1, engine=eng)   # sharded over the two ranks
        assert np.array_equal(hdw, o["hdw"]), "sharded Hamming weights differ from the golden ones"
        red = MIH.perform_MI_computation(sd, o["hdw"], CdsVar(paint=o["paint"], nclust=3), ncores=1,
                                         lr_save_path=os.path.join(outdir, "lr_links.tsv"), sr_save_path=os.path.join(outdir, "sr_links.tsv"),
                                         plt_folder=os.path.join(outdir, "PLOTS"), max_blk_sz=1000, lr_retain_links=1e5, engine=eng,
                                         perform_SR_analysis_only=("sr_only" in mode), sr_dist=(3000 if "sr_only" in mode else 20000),
                                         verbose=False, quirk_mode=1,   # LDW_QUIRK_INTENDED, like the single-process run of the test
                                         sr_tail=("dist" if "rows_stay" in mode else "gather"))
    if dist.get_rank() == 0:
        assert red is not None
        red.to_pickle(os.path.join(outdir, "red.pkl"))
    else:
        assert red is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
