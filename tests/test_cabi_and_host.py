"""CPU: the C-ABI library loads and exports what include/ldweaver_amd.h declares; host logic (blocks, RNG-driven
lr_links_approx, srp model, ARACNE and the small native helpers) against the oracle.  No GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pandas as pd
import pytest

import ldw_oracle as orc
from ldweaver_amd import _lib as L
from ldweaver_amd import mi as MI
from ldweaver_amd import srp
from ldweaver_amd.dist import block_cost, deal_blocks
from ldweaver_amd.engine import aracne
from ldweaver_amd.snpdat import SnpDat, encode_chars

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ldweaver_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ldw_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("ldw_ctx")
    assert len(declared) >= 30
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(L.declared_symbols()), declared ^ set(L.declared_symbols())
    assert lib.ldw_version() >= 100


def test_default_library_is_lean():
    """the library the package loads by default is the production build — no LDW_EXPERIMENTS kernels or switches
    (ldw_build_info bit 0 clear), under 8 MB; the experiments build is a separate file (make EXPERIMENTS=1, LDW_AMD_LIB)."""
    if os.environ.get("LDW_AMD_LIB"):
        pytest.skip("LDW_AMD_LIB points at another build")
    assert L.lib().ldw_build_info() == 0 and not L.has_experiments()
    assert os.path.getsize(L.LIB_PATH) < 8 * 1024 * 1024, os.path.getsize(L.LIB_PATH)
    syms = os.popen(f"nm -D --defined-only {L.LIB_PATH}").read()
    for name in ("launch_fused", "launch_hist"):   # (host launchers of experiment kernels; device kernels live in the fat binary)
        assert name not in syms, name


def test_host_trim_without_a_context(tmp_path):
    """ldw_host_trim(NULL): the tsv writers' process-wide buffer pool goes back to the OS; the writer works again afterwards."""
    cols = [np.arange(50_000, dtype=np.int32), np.linspace(0, 1, 50_000)]
    MI.append_table(str(tmp_path / "a.tsv"), cols)
    n = C.c_int64(-1)
    assert L.lib().ldw_host_trim(None, C.byref(n)) == L.LDW_OK and n.value > 0
    assert L.lib().ldw_host_trim(None, C.byref(n)) == L.LDW_OK and n.value == 0
    MI.append_table(str(tmp_path / "b.tsv"), cols)
    assert (tmp_path / "a.tsv").read_bytes() == (tmp_path / "b.tsv").read_bytes()


def test_no_cpu_fallback():
    """Without a GPU the context cannot be created and says so; with one the test is vacuous."""
    lib = L.lib()
    if lib.ldw_device_count() > 0:
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = lib.ldw_ctx_create(0, C.byref(ctx))
    assert rc == 4  # LDW_ERR_NOGPU
    assert b"no CPU fallback" in lib.ldw_last_error()
    from ldweaver_amd.engine import Engine
    with pytest.raises(L.LdwError):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ldweaver_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "ldw_oracle" not in txt and "c_oracle" not in txt and "oracle/" not in txt, f


def test_make_blocks_and_deal():
    b = MI.make_blocks(1268, 1000)
    assert b.tolist() == [list(x) for x in orc.make_blocks(1268, 1000)] == [[1, 1000, 1, 1000], [1, 1000, 1001, 1268], [1001, 1268, 1001, 1268]]
    b = MI.make_blocks(100000, 10000)
    assert len(b) == 55 and b[0].tolist() == [1, 10000, 1, 10000] and b[-1].tolist() == [90001, 100000, 90001, 100000]
    with pytest.raises(ValueError):
        MI.make_blocks(100, 0)
    cost = block_cost(b)
    assert cost[0] == int(10000 * 9999 // 2 * 3.3) and cost[1] == 10 ** 8      # a diagonal pair holds the dense short-range band: 1.65 off-diagonal ones
    assert block_cost(b, diag_factor=1.0)[0] == 10000 * 9999 // 2
    for world in (1, 2, 4, 8):
        parts = deal_blocks(b, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(55))
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() <= 1.08 * loads.mean() + 1
        assert all((np.diff(p) > 0).all() for p in parts if len(p) > 1)


def test_lr_links_approx_matches_oracle(sample):
    a = MI.lr_links_approx(sample["POS"], sample["g"], 20000)
    assert a == orc.lr_links_approx(sample["POS"], sample["g"], 20000) == float(sample["single_lr_approx"])


def test_snpdat_and_encoder():
    st = np.array([[0, 1, 4], [2, 2, 3]], dtype=np.uint8)
    sd = SnpDat.from_states(st, [5, 9], 100)
    assert sd.nsnp == 2 and sd.nseq == 3 and sd.r.tolist() == [3.0, 2.0]
    assert sd.uqe.tolist() == [[1, 1, 0, 0, 1], [0, 0, 1, 1, 0]]
    sd2 = SnpDat.from_onehots(sd.onehots(), [5, 9], 100)
    assert np.array_equal(sd2.states, st)
    with pytest.raises(ValueError):
        SnpDat.from_onehots([np.zeros((2, 3))] * 5, [5, 9], 100)
    assert encode_chars(np.frombuffer(b"AcgTn-RY*", dtype=np.uint8)).tolist() == [0, 1, 2, 3, 4, 4, 4, 4, 4]


def _toy_links(seed=0, n=4000):
    rng = np.random.default_rng(seed)
    pos1 = rng.integers(1, 3000, n).astype(float)
    ln = rng.integers(1, 400, n).astype(float)
    pos2 = pos1 + ln
    c1 = rng.integers(1, 3, n)
    c2 = np.where(rng.random(n) < 0.8, c1, 3 - c1)
    base = 0.05 * ln ** -0.3
    mi = base * rng.gamma(2.0, 0.5, n)
    return pd.DataFrame({"pos1": pos1, "pos2": pos2, "clust1": c1, "clust2": c2, "len": ln, "MI": mi})


def test_srp_model_matches_oracle():
    df = _toy_links()
    by_clust = [df[(df.clust1 == ci) | (df.clust2 == ci)] for ci in (1, 2)]
    fd, ofd = [], []
    red, chk = srp.merge_n_sort_sr_links(by_clust, 2, 20000, 1.0, fit_data=fd)
    ored, ochk = orc.merge_n_sort_sr_links([{k: d[k].to_numpy() for k in srp.COLS} for d in by_clust], 2, 20000, 1.0, fit_data=ofd)
    assert len(fd) == len(ofd) == 2                      # the maxvls table of c<i>_fit_data.rds (R/computePairwiseMI.R:422-439)
    for a, b in zip(fd, ofd):
        assert list(a.columns) == ["len", "max", "fit"] and np.array_equal(a["len"].to_numpy(), b["len"])
        np.testing.assert_allclose(a["max"].to_numpy(), b["max"], rtol=1e-12)
        np.testing.assert_allclose(a["fit"].to_numpy(), b["fit"], rtol=1e-10)
    assert len(red) == len(ored["MI"]) > 10 and len(chk) == len(ochk["MI"])
    for k in ("clust_c", "pos1", "pos2", "clust1", "clust2", "len", "MI"):
        assert np.array_equal(red[k].to_numpy(dtype=float), np.asarray(ored[k], dtype=float)), k
    np.testing.assert_allclose(red["srp_max"].to_numpy(), ored["srp_max"], rtol=1e-9)
    assert (red["srp_max"] > 1.0).all()
    with pytest.raises(ValueError):
        srp.merge_n_sort_sr_links(by_clust, 3, 20000, 1.0)


def test_aracne_native_matches_oracle():
    rng = np.random.default_rng(3)       # dense graph on 60 positions: plenty of triangles, some duplicated links
    p1 = rng.integers(1, 61, 700).astype(float) * 10
    p2 = rng.integers(1, 61, 700).astype(float) * 10
    keep = p1 != p2
    full = pd.DataFrame({"pos1": p1[keep], "pos2": p2[keep], "MI": rng.random(int(keep.sum()))})
    chk = full[full.MI > full.MI.quantile(0.5)]
    got = aracne(chk.pos1, chk.pos2, chk.MI, full.pos1, full.pos2, full.MI)
    ref = orc.run_aracne(chk.pos1.to_numpy(), chk.pos2.to_numpy(), chk.MI.to_numpy(), full.pos1.to_numpy(), full.pos2.to_numpy(),
                         full.MI.to_numpy())
    assert np.array_equal(got, ref) and (~ref).sum() > 0 and ref.sum() > 0
    # links with no neighbours stay TRUE; empty input
    assert aracne([1.], [2.], [0.3], [1.], [2.], [0.3]).tolist() == [True]
    assert aracne([], [], [], [], [], []).tolist() == []


def test_native_helper_twins():
    lib = L.lib()
    rng = np.random.default_rng(1)
    x = np.asfortranarray(rng.integers(1, 6, (40, 2)).astype(np.float64))
    y = np.array([2.0, 5.0])
    ret = np.zeros(40, dtype=np.uint8)
    L.check(lib.ldw_compare_to_row(L.ptr(x), 40, 2, L.ptr(y), 2, L.ptr(ret)))
    assert np.array_equal(ret.astype(bool), orc.compare_to_row(x, y))
    xs, ys = np.array([3.0, 9.0, 1.0]), np.array([1.0, 3.0, 3.0])
    out = np.zeros(3)
    L.check(lib.ldw_vec_pos_match(L.ptr(xs), 3, L.ptr(ys), 3, L.ptr(out)))
    assert out.tolist() == list(orc.vec_pos_match(xs, ys)) == [2.0, 0.0, 1.0]
    r = C.c_int(-1)
    a, b = np.array([0.5, 0.1]), np.array([0.3, 0.9])
    L.check(lib.ldw_compare_triplet(L.ptr(a), L.ptr(b), 2, 0.2, C.byref(r)))
    assert r.value == 0 == int(orc.compare_triplet(a, b, 0.2))
    L.check(lib.ldw_compare_triplet(L.ptr(a), L.ptr(b), 2, 0.4, C.byref(r)))
    assert r.value == 1
    A, B = np.array([5, 1, 3, 3], dtype=np.int32), np.array([3, 3, 7, 5], dtype=np.int32)
    o = np.zeros(4, dtype=np.int32)
    n = C.c_int64(0)
    L.check(lib.ldw_fast_intersect(L.ptr(A), 4, L.ptr(B), 4, L.ptr(o), C.byref(n)))
    assert o[:n.value].tolist() == orc.fast_intersect(A, B) == [3, 3, 5]
    assert lib.ldw_fast_intersect(None, 0, None, 0, None, None) == 1  # LDW_ERR_ARG with a message
    assert b"bad argument" in lib.ldw_last_error()


def test_append_table_format(tmp_path):
    p = tmp_path / "t.tsv"
    MI.append_table(str(p), [np.array([1, 2]), np.array([100000.0, 20000.0]), np.array([0.123456789012345678, 1e-5])])
    MI.append_table(str(p), [np.array([3]), np.array([5.0]), np.array([0.5])])
    assert p.read_text() == "1\t1e+05\t0.123456789012346\n2\t20000\t1e-05\n3\t5\t0.5\n"


def test_lr_links_approx_fast_path_equals_brute_force():
    """Above 2000 SNPs the host uses binary searches on the ascending POS instead of the reference's O(n^2/10) scan:
    same seeded sample, identical count (also across the origin)."""
    rng = np.random.default_rng(8)
    for n, g, sr in ((5000, 2_221_315, 20000), (3000, 60_001, 20000), (2500, 50_000, 12000)):
        POS = np.sort(rng.choice(g, n, replace=False) + 1)
        assert MI.lr_links_approx(POS, g, sr) == orc.lr_links_approx(POS, g, sr)


def test_near_mask_equals_the_references_site_filter():
    """perform_SR_analysis_only's kp_f / kp_t (R/computePairwiseMI.R:182-183), computed in strips."""
    rng = np.random.default_rng(5)
    g, sr = 100000.0, 40.0
    pf = np.sort(rng.integers(1, 100001, 900)).astype(np.float64)
    pt = np.sort(rng.integers(1, 100001, 1100)).astype(np.float64)
    kf, kt = MI._near_mask(pf, pt, g, sr)
    ln = np.abs(orc.circ_len(pt[None, :], pf[:, None], g))
    assert np.array_equal(kf, (ln < sr).any(axis=1)) and np.array_equal(kt, (ln < sr).any(axis=0))
    assert 0 < kf.sum() < len(kf)


def test_lr_links_approx_on_unsorted_positions_matches_oracle():
    """snp.dat$POS may be in any order (the reference imposes none, R/computePairwiseMI.R:94-97 works on whatever it gets): the host's
    binary-search shortcut only applies to ascending positions, the general scan must give the oracle's number."""
    rng = np.random.default_rng(5)
    POS = rng.permutation(np.sort(rng.choice(500_000, 2600, replace=False) + 1)).astype(np.int32)
    assert np.any(np.diff(POS) < 0)
    assert MI.lr_links_approx(POS, 500_000.0, 20000.0) == orc.lr_links_approx(POS, 500_000.0, 20000.0)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` without a launcher around it must start 2 ranks (torch.distributed.run as a CHILD process) instead of
    silently running one.  No GPU here: the ranks fail at their first device call, and that failure must come back
    as a non-zero exit code of bench.py itself — with no JSON line — after the launcher command has been announced on stderr."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--L", "2000", "--N", "200", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--no-extra-legs"], capture_output=True, text=True, timeout=600, env=env)
    assert "starting 2 ranks" in p.stderr and "--nproc-per-node=2" in p.stderr and "--master-addr 127.0.0.1" in p.stderr, p.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_positions_to_snp_index_any_order():
    """lr.analyse_long_range_links maps the positions of sr_links.tsv back to SNP indices with POS in any order."""
    from ldweaver_amd.lr import positions_to_snp_index
    rng = np.random.default_rng(3)
    POS = rng.permutation(np.arange(10, 5000, 7))
    pick = rng.integers(0, len(POS), 300)
    assert np.array_equal(positions_to_snp_index(POS, POS[pick]), pick)
    assert np.array_equal(positions_to_snp_index(np.sort(POS), np.sort(POS)[pick]), pick)
    assert len(positions_to_snp_index(POS, np.zeros(0, dtype=POS.dtype))) == 0
    dup = np.array([5, 9, 5, 7])
    assert positions_to_snp_index(dup, np.array([5, 7])).tolist() == [0, 3]          # first holder of a repeated position
    with pytest.raises(ValueError):
        positions_to_snp_index(POS, np.array([11]))


def test_cpu_share_and_thread_pool_limits(monkeypatch):
    """cpushare: the CPUs the cgroup / affinity mask grants (never more than the visible count), and environment defaults that leave the
    user's own settings alone."""
    import os
    from ldweaver_amd.cpushare import cpu_share, limit_thread_pools
    n = cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    monkeypatch.setenv("OPENBLAS_NUM_THREADS", "3")
    assert limit_thread_pools(5) == 5
    assert os.environ["OMP_NUM_THREADS"] == "5" and os.environ["OPENBLAS_NUM_THREADS"] == "3"


def test_r_shim_type_checks_and_links_against_the_library(tmp_path):
    """The image has no R, so the shim (r_shim/ldweaver_amd_shim.c) had never met a compiler: gcc type-checks it here against include/ldweaver_amd.h and a
    DECLARATION-ONLY stand-in for the part of R's C API it uses (tests/r_api_mock/: test infrastructure, not R) — wrong arities or pointer types in a
    call into the library are errors — and every ldw_* symbol the object needs must be exported by the built library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or shutil.which("nm") is None:
        pytest.skip("no gcc / nm")
    src = os.path.join(ROOT, "r_shim", "ldweaver_amd_shim.c")
    inc = ["-I", os.path.join(ROOT, "tests", "r_api_mock"), "-I", os.path.join(ROOT, "include")]
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-cast-function-type", "-Werror"] + inc + [src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    obj = str(tmp_path / "shim.o")
    r = subprocess.run(["gcc", "-std=c11", "-c", "-fPIC"] + inc + [src, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    needs = {ln.split()[-1] for ln in subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout.splitlines() if ln.split()[-1].startswith("ldw_")}
    assert len(needs) >= 30, sorted(needs)
    lib = L.lib()
    missing = [n for n in sorted(needs) if not hasattr(lib, n)]
    assert not missing, missing
    # the six routines of the reference's registration table (src/RcppExports.cpp:154-160) are defined by the object
    defined = {ln.split()[-1] for ln in subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True).stdout.splitlines() if ln.strip()}
    for name in ("_LDWeaver_ACGTN2num", "_LDWeaver_fastHadamard", "_LDWeaver_compareToRow", "R_init_ldweaver_amd_shim"):
        assert name in defined, name


def test_r_shim_is_consistent_with_itself_and_the_header():
    """The R shim cannot be compiled here (no R headers in the image), so it is checked statically: every `.Call("ldwamd_*", ...)` of
    r_shim/ldweaver_amd.R names a routine registered in the shim's R_CallMethodDef table with the number of arguments the call passes; every
    registered routine is defined with that many SEXP parameters; every `ldw_*` function the shim calls is declared in include/ldweaver_amd.h
    and called with the declared number of arguments; the reference's own six `.Call` symbols are registered with the reference's arities
    (src/RcppExports.cpp:154-160)."""
    c_src = open(os.path.join(ROOT, "r_shim", "ldweaver_amd_shim.c")).read()
    r_src = open(os.path.join(ROOT, "r_shim", "ldweaver_amd.R")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ldweaver_amd.h")).read(), flags=re.S)
    c_code = re.sub(r"/\*.*?\*/", "", c_src, flags=re.S)

    def split_args(s):
        out, depth, cur, in_str = [], 0, "", None
        for ch in s:
            if in_str:
                cur += ch
                if ch == in_str:
                    in_str = None
                continue
            if ch in "\"'":
                in_str = ch
                cur += ch
            elif ch in "([{":
                depth += 1
                cur += ch
            elif ch in ")]}":
                depth -= 1
                cur += ch
            elif ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur)
        return out

    def call_args(src, start):
        """argument list of the call whose '(' is at src[start]"""
        depth, i = 0, start
        while True:
            depth += src[i] == "("
            depth -= src[i] == ")"
            if depth == 0:
                return split_args(src[start + 1:i])
            i += 1

    table = {m.group(1): int(m.group(3)) for m in re.finditer(r'\{"(\w+)",\s*\(DL_FUNC\)\s*&(\w+),\s*(\d+)\}', c_code)}
    assert len(table) >= 20
    for name, arity in table.items():
        m = re.search(r"\bSEXP\s+" + re.escape(name if not name.startswith("_LDWeaver_") else name) + r"\s*\(", c_code)
        if m is None:      # (the reference-named entries point at differently named adapters)
            target = re.search(r'\{"' + re.escape(name) + r'",\s*\(DL_FUNC\)\s*&(\w+)', c_code).group(1)
            m = re.search(r"\bSEXP\s+" + re.escape(target) + r"\s*\(", c_code)
        assert m, name
        params = call_args(c_code, m.end() - 1)
        n = 0 if [p.strip() for p in params] in ([], ["void"]) else len(params)
        assert n == arity, (name, n, arity)
    for m in re.finditer(r'\.Call\("(\w+)"', r_src):
        name = m.group(1)
        assert name in table, f".Call of an unregistered routine {name}"
        nargs = len(call_args(r_src, r_src.index("(", m.start()))) - 1
        assert nargs == table[name], (name, nargs, table[name])
    want = dict(_LDWeaver_ACGTN2num=3, _LDWeaver_fastHadamard=9, _LDWeaver_compareToRow=2, _LDWeaver_vecPosMatch=2, _LDWeaver_compareTriplet=3,
                _LDWeaver_fast_intersect=2)
    for k, v in want.items():
        assert table.get(k) == v, (k, table.get(k))
    declared = {}
    for m in re.finditer(r"\b(ldw_[a-z0-9_]+)\s*\(", hdr):
        args = call_args(hdr, m.end() - 1)
        declared[m.group(1)] = 0 if [a.strip() for a in args] == ["void"] else len(args)
    for m in re.finditer(r"\b(ldw_[a-z0-9_]+)\s*\(", c_code):
        name = m.group(1)
        assert name in declared, f"the shim calls {name}, which the header does not declare"
        n = len(call_args(c_code, m.end() - 1))
        assert n == declared[name], (name, n, declared[name])


def test_new_entry_points_refuse_bad_arguments_without_a_gpu():
    """Argument checks of the r05 entry points that need no device: null / out-of-range arguments come back as LDW_ERR_ARG with a message,
    never as a crash."""
    lib = L.lib()
    blocks = np.ascontiguousarray(MI.make_blocks(3000, 1000), dtype=np.int32)
    owner = np.zeros(len(blocks), dtype=np.int32)
    assert lib.ldw_deal_blocks(None, len(blocks), 2, L.ptr(owner)) == L.LDW_ERR_ARG
    assert lib.ldw_deal_blocks(L.ptr(blocks), len(blocks), 0, L.ptr(owner)) == L.LDW_ERR_ARG
    assert lib.ldw_deal_blocks(L.ptr(blocks), 0, 2, L.ptr(owner)) == L.LDW_ERR_ARG
    p = L.MIParams(20000.0, 1e6, 1.0, 0, 0, 1, 0)
    assert lib.ldw_mi_all_pairs_multi(None, 2, L.ptr(blocks), len(blocks), C.byref(p), None, None) == L.LDW_ERR_ARG
    arr = (C.c_void_p * 2)(None, None)
    assert lib.ldw_mi_all_pairs_multi(arr, 2, L.ptr(blocks), len(blocks), C.byref(p), None, None) == L.LDW_ERR_ARG
    assert b"context 0 is null" in lib.ldw_last_error()
    assert lib.ldw_mi_all_pairs_multi(arr, 65, L.ptr(blocks), len(blocks), C.byref(p), None, None) == L.LDW_ERR_ARG
    assert lib.ldw_hamming_weights_multi(None, 1, 10, L.ptr(np.zeros(4))) == L.LDW_ERR_ARG
    assert lib.ldw_lr_stream_end(None, None, None, None) == L.LDW_ERR_ARG
    assert lib.ldw_tsv_join(None) == L.LDW_ERR_ARG
    assert lib.ldw_overflow_report(None, L.ptr(np.zeros(4, dtype=np.int64))) == L.LDW_ERR_ARG
    assert lib.ldw_build_info() in (0, 1)
    # r05, the short-range model over ranks: no context -> LDW_ERR_ARG, never a crash
    z8, zi = np.zeros(8), np.zeros(8, dtype=np.int64)
    n = C.c_int64(0)
    assert lib.ldw_sr_tail_extract(None, 1, 4, L.ptr(z8), L.ptr(zi), None, 0, 0, C.byref(n)) == L.LDW_ERR_ARG
    assert lib.ldw_sr_quantiles_merge(None, 1, 4, 0.95, 1, None, None, L.ptr(zi), 0, L.ptr(z8), L.ptr(z8), None) == L.LDW_ERR_ARG
    assert lib.ldw_sr_excess_stats_blocks(None, 1, 4, L.ptr(z8), 1, L.ptr(zi), L.ptr(z8)) == L.LDW_ERR_ARG
    assert lib.ldw_sr_pool_build(None, 0.1, C.byref(n)) == L.LDW_ERR_ARG
    assert lib.ldw_sr_reduced_import(None, 0, None, None, None, 0, None, None, None) == L.LDW_ERR_ARG
    assert b"null context" in lib.ldw_last_error()
