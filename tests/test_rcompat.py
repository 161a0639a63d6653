"""CPU: base-R behaviours restated in ldweaver_amd/rcompat.py, pinned on known answers from R itself."""
import numpy as np

import ldw_oracle as orc
from ldweaver_amd import rcompat


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
