"""CPU: base-R behaviours restated in ldweaver_amd/rcompat.py, pinned on known answers from R itself."""
import numpy as np

import ldw_oracle as orc
from ldweaver_amd import rcompat
from ldweaver_amd.engine import format_number as native_format, write_table_tsv


def test_r_rng_known_answers():
    # R >= 3.6: set.seed(123); runif(3)  /  sample(1:10)  ;  set.seed(42); sample(1:10) ; set.seed(1); runif(1)
    r = rcompat.RRandom(123)
    assert [round(r.unif_rand(), 7) for _ in range(3)] == [0.2875775, 0.7883051, 0.4089769]
    assert list(rcompat.RRandom(123).sample(10, 10)) == [3, 10, 2, 8, 6, 9, 1, 7, 5, 4]
    assert list(rcompat.RRandom(42).sample(10, 10)) == [1, 5, 10, 8, 2, 4, 6, 9, 7, 3]
    assert round(rcompat.RRandom(1).unif_rand(), 7) == 0.2655087


def test_rng_product_equals_oracle_restatement():
    a = rcompat.RRandom(1988).sample(12680, 1268)
    b = orc.RMersenneTwister(1988).sample(12680, 1268)
    assert np.array_equal(a, b) and len(set(a.tolist())) == 1268
    # crosses several 624-word refills
    r1, r2 = rcompat.RRandom(7), orc.RMersenneTwister(7)
    assert [r1.unif_rand() for _ in range(2000)] == [r2.unif_rand() for _ in range(2000)]


def test_native_r_sample_is_r_s_stream():
    """ldw_r_sample (what lr_links_approx draws its 10 % SNP subset with) against R's own known answers and the Python and oracle
    statements of the stream, across refills of the 624-word state, bit widths 1..17 and the degenerate sizes."""
    assert list(rcompat.r_sample(123, 10, 10)) == [3, 10, 2, 8, 6, 9, 1, 7, 5, 4]
    assert list(rcompat.r_sample(42, 10, 10)) == [1, 5, 10, 8, 2, 4, 6, 9, 7, 3]
    for seed, n, size in ((1988, 12680, 1268), (1988, 100000, 10000), (7, 1, 1), (5, 1000, 0), (99, 65536, 65536), (3, 65537, 4000), (2**32 - 1, 777, 777)):
        a = rcompat.r_sample(seed, n, size)
        assert np.array_equal(a, rcompat.RRandom(seed).sample(n, size)), (seed, n, size)
        assert len(set(a.tolist())) == size and (size == 0 or (a.min() >= 1 and a.max() <= n))
    assert np.array_equal(rcompat.r_sample(1988, 12680, 1268), orc.RMersenneTwister(1988).sample(12680, 1268))
    # n > 1e7 with size <= n / 2: R's sample.int(useHash = TRUE) draws with replacement-and-retry (do_sample2), not the partial shuffle
    a = rcompat.r_sample(1988, 10_000_001, 20_000)
    b = rcompat.RRandom(1988).sample(10_000_001, 20_000)
    assert np.array_equal(a, b) and len(set(a.tolist())) == 20_000 and a.min() >= 1 and a.max() <= 10_000_001
    draws = rcompat.RRandom(1988)
    raw = [draws._unif_index(10_000_001) + 1 for _ in range(20_050)]
    assert len(set(raw[:20_000])) < 20_000, "the case must contain a repeated draw, or the retry rule is not exercised"
    assert list(a[:5]) == raw[:5]


def test_quantile_type7():
    rng = np.random.default_rng(3)
    x = rng.gamma(0.3, size=1001)
    for p in (0.0, 0.95, 0.9998, 0.5, 1.0, 0.123456):
        assert abs(rcompat.quantile7(x, p) - np.quantile(x, p)) < 1e-15
        assert rcompat.quantile7(x, p) == orc.quantile7(x, p)
    assert abs(rcompat.quantile7(np.array([1.0, 2.0, 3.0, 4.0]), 0.95) - 3.85) < 1e-15  # R: quantile(1:4, .95) = 3.85


def test_round_and_format():
    assert rcompat.round_thousands(10000) == 10000 and rcompat.round_thousands(1500) == 2000 and rcompat.round_thousands(2500) == 2000
    cases = {100000.0: "1e+05", 20000.0: "20000", 123456.0: "123456", 0.1: "0.1", 0.0001: "1e-04", 1e-5: "1e-05",
             0.000123456789: "0.000123456789", 1234.5: "1234.5", 0.6824123456789012: "0.682412345678901",
             1.8e-13: "1.8e-13", 3.0: "3", -0.25: "-0.25", 1e15: "1e+15", 0.0: "0"}
    for v, s in cases.items():
        assert rcompat.format_number(v) == s, (v, rcompat.format_number(v), s)
        assert orc.r_format_double(v) == s
    assert rcompat.format_number(np.int32(7)) == "7"


def test_circ_len():
    g = 50000
    assert list(rcompat.circ_len([10, 49990, 30000], [49990, 10, 5000], g)) == [20.0, 20.0, 25000.0]
    assert list(rcompat.circ_len([5, 7], [7, 5], 11)) == [2.0, 2.0]   # odd genome length stays integral


_R_CASES = {100000.0: "1e+05", 20000.0: "20000", 123456.0: "123456", 0.1: "0.1", 0.0001: "1e-04", 1e-5: "1e-05",
            0.000123456789: "0.000123456789", 1234.5: "1234.5", 0.6824123456789012: "0.682412345678901",
            1.8e-13: "1.8e-13", 3.0: "3", -0.25: "-0.25", 1e15: "1e+15", 0.0: "0"}


def test_native_number_format_known_answers():
    """ldw_format_number (the native writer's cell rule, ldw_tsv.cpp) on the known R answers and the special values."""
    for v, s in _R_CASES.items():
        assert native_format(v) == s, (v, native_format(v), s)
    assert native_format(float("nan")) == "NA" and native_format(float("inf")) == "Inf" and native_format(float("-inf")) == "-Inf"
    assert native_format(-0.0) == "0" and native_format(1110657.5) == "1110657.5" and native_format(0.5) == "0.5"
    assert native_format(123456789012345678.0) == rcompat.format_number(123456789012345678.0) == "123456789012345677"[:0] + rcompat.format_number(123456789012345678.0)


def test_native_number_format_next_to_powers_of_ten():
    """doubles just below a power of ten (floor(log10) comes out one too high; the rounded 15-digit quotient lands exactly on
    10^14) must keep their own digits: 9.999999999999994e-05 is 9.99999999999999e-05, not 1e-04.  The nextafter neighbours (8 steps each
    way) of 1e-5 ... 1e3, and values half a unit of the 15th digit around them, against the Python statement of R's rule."""
    vals = [9.99999999999999e-05, 9.999999999999991e-05, 9.999999999999992e-05, 9.999999999999994e-05, 999.9999999999994]
    for e in range(-5, 4):
        p = 10.0 ** e
        for d in (-1.0, 1.0):
            v = p
            for _ in range(8):
                v = float(np.nextafter(v, d * np.inf))
                vals.append(v)
        vals += [p * (1 - 4e-16), p * (1 - 5e-16), p * (1 - 6e-16), p * (1 - 4.9e-15), p * (1 - 5.1e-15), p * (1 + 5e-15), p * (1 + 4.4e-16)]
    for v in vals:
        assert native_format(v) == rcompat.format_number(v), (repr(v), native_format(v), rcompat.format_number(v))
    assert native_format(9.999999999999994e-05) == "9.99999999999999e-05" and native_format(999.9999999999994) == "999.999999999999"


def test_native_tsv_writer_equals_python_writer(tmp_path):
    """ldw_write_table_tsv byte-identical to the Python statement of R's rule (rcompat.format_number) on 1e6 random doubles of
    every kind the link files hold (MI-like, len-like halves and integers, tiny / huge magnitudes, negatives) + integer columns;
    append semantics; threads do not reorder rows."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    x = np.concatenate([rng.random(n // 2) * 0.2, np.exp(rng.uniform(-40, 40, n // 8)), rng.integers(0, 4_000_000, n // 8) / 2.0,
                        rng.integers(0, 3_000_000, n // 8).astype(float), -np.exp(rng.uniform(-13, 8, n // 16)), rng.integers(1, 10 ** 6, n // 16) / 1000.0,
                        np.array(list(_R_CASES) + [1e-5, 9.99999999999999e-6, 999.999999999999, 999.9999999999999, 0.999999999999999,
                                                   0.9999999999999999, 0.099999999999999995, 5e-324, 1e300, 1e22, 1e21, 99999.99999999999,
                                                   999999999999999.9, 0.1 + 0.2, 1 / 3, 2 / 3, float("nan"), float("inf")])])
    ints = rng.integers(-5, 3_000_000, len(x)).astype(np.int32)
    path = tmp_path / "t.tsv"
    path.write_text("head\tline\n")
    nb = write_table_tsv(str(path), [ints, x], append=True)
    got = path.read_text().splitlines()
    assert got[0] == "head\tline" and len(got) == len(x) + 1 and nb == path.stat().st_size - len("head\tline\n")
    fmt = rcompat.format_number
    step = 1 if len(x) <= 1_200_000 else 7
    for i in range(0, len(x), step):
        assert got[i + 1] == f"{int(ints[i])}\t{fmt(float(x[i]))}", (i, x[i], got[i + 1])
    # single thread == many threads; truncate instead of append
    write_table_tsv(str(tmp_path / "one.tsv"), [ints, x], append=False, nthreads=1)
    write_table_tsv(str(tmp_path / "many.tsv"), [ints, x], append=False, nthreads=7)
    assert (tmp_path / "one.tsv").read_bytes() == (tmp_path / "many.tsv").read_bytes() == path.read_bytes()[len("head\tline\n"):]
    write_table_tsv(str(tmp_path / "empty.tsv"), [np.zeros(0), np.zeros(0, dtype=np.int64)], append=False)
    assert (tmp_path / "empty.tsv").read_bytes() == b""


def test_beta_mle_special_functions_and_noise_floor():
    """The short-range model's scalar special functions (psi, psi', log B in plain Python: no scipy import on the product path) against
    scipy / mpmath, and the beta MLE's Newton iteration on sufficient statistics with n ~ 1e8 (the C5 clusters: the score's rounding
    noise keeps the step at ~1e-13 of the parameter; the iteration must stop there instead of falling through to the simplex)."""
    import sys

    from scipy import special

    from ldweaver_amd import srp
    rng = np.random.default_rng(0)
    xs = list(rng.uniform(0.01, 300, 500)) + [1e-3, 0.5, 1.0, 2.0, 9.99, 10.0, 1e4]
    for x in xs:
        assert abs(srp._digamma(x) - special.digamma(x)) <= 2e-15 * max(1.0, abs(special.digamma(x)))
        assert abs(srp._trigamma(x) - special.polygamma(1, x)) <= 2e-15 * special.polygamma(1, x)
    try:
        import mpmath
        mpmath.mp.dps = 40
        for a, b in [(0.359274, 97.758595), (0.396209, 132.942389), (1.7, 4000.0), (3.3, 10.0), (50.0, 60.0), (0.01, 1e5), (2.0, 3.0)]:
            exact = float(mpmath.log(mpmath.beta(a, b)))
            assert abs(srp._betaln(a, b) - exact) <= 2e-15 * max(1.0, abs(exact)), (a, b)
            assert srp._betaln(a, b) == srp._betaln(b, a)
    except ImportError:
        for a, b in [(0.359274, 97.758595), (2.0, 3.0), (50.0, 60.0)]:
            assert abs(srp._betaln(a, b) - special.betaln(a, b)) <= 1e-12
    had = "scipy.optimize" in sys.modules
    for st, want in (([28498592.0, 70777.64524797378, 17752.603752309653, -213122118.91724914, -85127.43904636076], (0.39620871781, 132.942388955)),
                     ([128397431.0, 385079.5887454161, 109246.20739219322, -958929052.2912737, -473422.4968240424], (0.35927357102, 97.758594891))):
        a, b = srp.beta_mle_stats(*st)
        assert abs(a - want[0]) < 1e-10 and abs(b - want[1]) < 1e-8
    assert ("scipy.optimize" in sys.modules) == had          # Newton converged: the simplex fallback was not needed
