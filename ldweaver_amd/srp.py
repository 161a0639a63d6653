"""Short-range p-value model and link merge: host logic of ``mergeNsort_sr_links``
(R/computePairwiseMI.R:400-495).  Runs on the host like the reference (pandas table work plus a two-parameter
optimiser) in ``merge_n_sort_sr_links``; ``merge_n_sort_sr_links_device`` is the same model with the O(#sr links) work
done on the device-resident link table (csrc/ldw_srp.hip), which is what ``perform_MI_computation`` uses.

Reproduced quirks: Q5 ``mean_dist[len]`` is indexed by the VALUE of ``len`` (:448) and Q6 ``srp_max`` is
a natural-log tail probability (:453).  The plot / RDS side outputs (:430-440) are not produced.
"""
from __future__ import annotations

import math

import numpy as np
import pandas as pd

COLS = ["pos1", "pos2", "clust1", "clust2", "len", "MI"]


def _digamma(x: float) -> float:
    """psi(x), x > 0: recurrence up to x >= 10, then the asymptotic series (truncation < 1e-17).  The device path of the short-range
    model needs psi, psi' and log B for a handful of scalars; importing scipy.special for them cost more wall clock than the whole
    model on the device (0.15-0.2 s of a 0.45 s job)."""
    acc = 0.0
    while x < 10.0:
        acc -= 1.0 / x
        x += 1.0
    f = 1.0 / (x * x)
    return acc + math.log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12.0))))))


def _trigamma(x: float) -> float:
    """psi'(x), x > 0 (same scheme)."""
    acc = 0.0
    while x < 10.0:
        acc += 1.0 / (x * x)
        x += 1.0
    f = 1.0 / (x * x)
    return acc + 1.0 / x + 0.5 * f + (1.0 / x) * f * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30 - f * (5.0 / 66 - f * (691.0 / 2730 - f * 7.0 / 6))))))


def _betaln(a: float, b: float) -> float:
    """log B(a, b).  lgamma(b) - lgamma(a + b) cancels ~350 against ~350 for the shapes this model meets (a ~ 0.4, b ~ 100: 1e-13 off);
    for the larger argument >= 10 the difference is taken from Stirling's series term by term instead:
    lgamma(b) - lgamma(a + b) = -(b - 1/2) log1p(a / b) - a ln(a + b) + a + S(b) - S(a + b),  S(z) = 1/(12 z) - 1/(360 z^3) + 1/(1260 z^5) - ..."""
    if a > b:
        a, b = b, a
    if b < 10.0:
        return math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)

    def stirling_tail(z):
        f = 1.0 / (z * z)
        return (1.0 / z) * (1.0 / 12 - f * (1.0 / 360 - f * (1.0 / 1260 - f * (1.0 / 1680 - f * (1.0 / 1188 - f * 691.0 / 360360)))))

    diff = -(b - 0.5) * math.log1p(a / b) - a * math.log(a + b) + a + stirling_tail(b) - stirling_tail(a + b)
    return math.lgamma(a) + diff


def beta_mle_stats(n: float, sx: float, sxx: float, slx: float, sl1x: float):
    """coef(fitdistrplus::fitdist(x, "beta")) from the sufficient statistics n, sum x, sum x^2, sum log x, sum log(1-x):
    the maximiser of the (strictly concave) beta log-likelihood, found by damped Newton steps on the score equations
    from fitdistrplus' moment start.  The reference reaches the same point with optim's Nelder-Mead to reltol 1e-8; the
    simplex is kept only as a fallback for starts Newton cannot use."""
    if n < 2:  # fitdistrplus::fitdist stops the same way; happens when the fit of (:428) is NaN or no link exceeds it
        raise ValueError("fitdist: data must be a numeric vector of length greater than 1 (no short-range link exceeds the fitted decay)")
    m = sx / n
    v = sxx / n - m * m                      # (n-1)/n * var(x)
    aux = m * (1 - m) / v - 1

    def ll(a, b):
        if not (a > 0 and b > 0):
            return -np.inf
        return (a - 1) * slx + (b - 1) * sl1x - n * _betaln(a, b)

    def newton(a, b, iters):
        cur = ll(a, b)
        for _ in range(iters):
            ga = n * (_digamma(a + b) - _digamma(a)) + slx
            gb = n * (_digamma(a + b) - _digamma(b)) + sl1x
            tab = _trigamma(a + b)
            haa, hbb, hab = n * (tab - _trigamma(a)), n * (tab - _trigamma(b)), n * tab
            det = haa * hbb - hab * hab
            da, db = (hbb * ga - hab * gb) / det, (haa * gb - hab * ga) / det
            if not (np.isfinite(da) and np.isfinite(db)):
                return a, b, False
            t = 1.0
            while t > 1e-10 and not ll(a - t * da, b - t * db) >= cur - 1e-12 * abs(cur):   # damping: stay feasible, do not go downhill
                t *= 0.5
            if t <= 1e-10:
                return a, b, abs(da) < 1e-9 * a and abs(db) < 1e-9 * b
            a, b = a - t * da, b - t * db
            cur = ll(a, b)
            # the score is a difference of terms ~ n * psi: with n ~ 1e8 its rounding noise moves the Newton step by ~1e-13 of the parameter and the
            # iteration jitters at that level for ever (C5, cluster 3: 2e-13 .. 7e-13 — it fell through to the simplex, 55 ms + scipy's import).
            # Full steps below 2e-12 relative are convergence; the maximiser is determined to ~1e-12 either way
            if t == 1.0 and abs(da) < 2e-12 * a and abs(db) < 2e-12 * b:
                return a, b, True
        return a, b, False

    a0, b0 = m * aux, (1 - m) * aux
    if a0 > 0 and b0 > 0 and np.isfinite(a0) and np.isfinite(b0):
        a, b, ok = newton(float(a0), float(b0), 100)
        if ok:
            return a, b
    from scipy import optimize   # (the fallback only: starts Newton cannot use)
    res = optimize.minimize(lambda p: -ll(p[0], p[1]), np.array([a0, b0]), method="Nelder-Mead",
                            options=dict(xatol=1e-10, fatol=1e-12, maxiter=5000, maxfev=10000))
    a, b, _ = newton(float(res.x[0]), float(res.x[1]), 50)
    return a, b


def beta_mle(x: np.ndarray):
    x = np.asarray(x, dtype=np.float64)
    return beta_mle_stats(x.size, float(np.sum(x)), float(np.sum(x * x)), float(np.sum(np.log(x))), float(np.sum(np.log1p(-x))))


def _beta_cf(a, b, x):
    """Continued fraction of the incomplete beta function (modified Lentz), vectorised over x."""
    tiny = 1e-300
    qab, qap, qam = a + b, a + 1.0, a - 1.0
    c = np.ones_like(x)
    d = 1.0 - qab * x / qap
    d = 1.0 / np.where(np.abs(d) < tiny, tiny, d)
    h = d.copy()
    live = np.ones(x.shape, dtype=bool)
    for m in range(1, 1001):
        m2 = 2.0 * m
        for aa in (m * (b - m) * x / ((qam + m2) * (a + m2)), -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2))):
            d = 1.0 + aa * d
            d = 1.0 / np.where(np.abs(d) < tiny, tiny, d)
            c = 1.0 + aa / c
            c = np.where(np.abs(c) < tiny, tiny, c)
            delta = d * c
            h = np.where(live, h * delta, h)
        live &= np.abs(delta - 1.0) >= 4e-16
        if not live.any():
            break
    return h


def neg_log_beta_sf(x, a: float, b: float) -> np.ndarray:
    """-pbeta(x, a, b, lower.tail = FALSE, log.p = TRUE)  (R/computePairwiseMI.R:453): the tail stays in log space, so
    p-values far below the smallest double keep their -log (scipy's logsf returns inf there).  Same evaluation as the
    device kernel (csrc/ldw_srp.hip)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros(x.shape)
    out[x >= 1.0] = np.inf
    ok = (x > 0.0) & (x < 1.0)
    xo = x[ok]
    lfront = a * np.log(xo) + b * np.log1p(-xo) - _betaln(a, b)
    low = xo < (a + 1.0) / (a + b + 2.0)
    res = np.empty(xo.shape)
    if low.any():
        res[low] = -np.log1p(-np.exp(lfront[low]) * _beta_cf(a, b, xo[low]) / a)
    if (~low).any():
        res[~low] = -(lfront[~low] + np.log(_beta_cf(b, a, 1.0 - xo[~low]) / b))
    out[ok] = res
    return out


def fit_decay(ulen: np.ndarray, maxvls: np.ndarray) -> np.ndarray:
    """mean_dist = exp(fitted(fastLm(log(maxvls) ~ log(len))))  (R/computePairwiseMI.R:428-429)."""
    X = np.column_stack([np.log(ulen), np.ones(len(ulen))])
    coef, *_ = np.linalg.lstsq(X, np.log(maxvls), rcond=None)
    return np.exp(X @ coef)


def stable_argsort_desc(v: np.ndarray) -> np.ndarray:
    """order(-v) of R (stable): numpy's default argsort (vectorised quicksort: 5x faster than its stable kinds at 250 000 doubles) is
    stable wherever no two values are equal, which is the rule for p-values; the stable sort runs only when a tie exists."""
    neg = -np.asarray(v)
    idx = np.argsort(neg)
    s = neg[idx]
    if len(s) > 1 and bool(np.any(s[1:] == s[:-1])) or bool(np.isnan(s).any()):
        return np.argsort(neg, kind="stable")
    return idx


def merge_n_sort_sr_links_device(eng, nclust: int, sr_dist: float, srp_cutoff: float, POS, paint, g, run_aracne=True, order_links=False,
                                 block_rows=None):
    """mergeNsort_sr_links + runARACNE with the link table left on the device by ``eng.mi_all_pairs``: the O(#links)
    work (per-length quantiles, excess statistics, p-values, de-duplication, ARACNE) runs in HBM, the host keeps the
    least-squares fit and the beta MLE.  Returns sr_links_red (same rows, order and columns as the host path) with the
    ARACNE column filled, and a dict of side results.

    ``block_rows`` (r05): short-range rows of every block of the pass that left the table, in table order.  The excess statistics are then
    summed per reference block and over the blocks in that order (``ldw_sr_excess_stats_blocks``) — the order the model over ranks uses
    (``dist_srp.merge_n_sort_sr_links_dist``), so that a job's sr_links.tsv is the same file, byte for byte, on one GPU, with the table gathered
    from several, and with the rows left on their ranks.  Without it: strips of the whole table (a table of unknown block structure)."""
    if int(np.max(paint)) > nclust or int(np.min(paint)) < 1:
        raise ValueError("Cluster mismatch detected, stopping!")
    qlo, qhi, cnt = eng.sr_len_quantiles(nclust, sr_dist, 0.95)
    S = qlo.shape[1]
    md = np.full((nclust, S), np.nan)
    lens = np.arange(1, S + 1, dtype=np.float64)
    fit_data = []                                       # the maxvls table saved as c<i>_fit_data.rds (:422-439): len, max, fit
    for ci in range(nclust):
        has = cnt[ci] > 0
        n = cnt[ci][has].astype(np.float64)
        index = 1 + (n - 1) * 0.95                      # quantile type 7 (stats::quantile)
        h = index - np.floor(index)
        lo, hi = qlo[ci][has], qhi[ci][has]
        maxvls = np.where((h > 0) & (hi != lo), (1 - h) * lo + h * hi, lo)
        mean_dist = fit_decay(lens[has], maxvls)
        md[ci, :len(mean_dist)] = mean_dist             # looked up by the VALUE of len (Q5)
        fit_data.append(pd.DataFrame({"len": lens[has], "max": maxvls, "fit": mean_dist}))
    if block_rows is not None and int(np.sum(block_rows)) == eng.links_count(0):
        stats = np.zeros((nclust, 5))
        for part in eng.sr_excess_stats_blocks(md, block_rows):   # in block order
            stats += part
    else:
        stats = eng.sr_excess_stats(md)
    shape = np.empty((nclust, 3))
    for ci in range(nclust):
        a_, b_ = beta_mle_stats(*stats[ci])
        shape[ci] = a_, b_, _betaln(a_, b_)
    n_red, n_pool, min_mi = eng.sr_pvalues(md, shape, srp_cutoff)
    red = eng.sr_reduced()
    flags = eng.aracne_device() if (run_aracne and n_red) else np.ones(n_red, dtype=bool)
    # reference row order: per cluster the links inside one cluster, then the cross-cluster links in order of first
    # appearance (R/computePairwiseMI.R:470-486)
    key_cl = np.where(red["dup"], red["first_clust"], red["clust_c"])
    row = np.asarray(red["row"], dtype=np.int64)
    if n_red and int(row.max()) < (1 << 48) and int(key_cl.max()) < (1 << 14):
        # one 64-bit key (dup | cluster | row) instead of a three-key lexsort: a third of the time at 250 000 rows
        order = np.argsort((np.asarray(red["dup"], dtype=np.int64) << 62) | (key_cl.astype(np.int64) << 48) | row)   # (keys are unique: rows are)
    else:
        order = np.lexsort((row, key_cl, red["dup"]))
    if order_links:
        # sr_links_red[order(-srp_max)] (R/computePairwiseMI.R:126; order() is stable) folded into the same gather: the columns are
        # permuted once, as arrays, instead of once here and once more as a DataFrame
        order = order[stable_argsort_desc(red["srp_max"][order])]
    return {k: v[order] for k, v in red.items()}, flags[order], dict(mean_dist=md, shape=shape, stats=stats, n_pool=n_pool,
                                                                     min_mi=min_mi, counts=cnt, fit_data=fit_data)


def merge_n_sort_sr_links(sr_links: list, nclust: int, sr_dist: float, srp_cutoff: float, fit_data: list | None = None):
    """sr_links: list (one per cluster) of DataFrames with COLS.  Returns (sr_links_red, sr_links_ARACNE_check); the per-cluster
    ``maxvls`` tables (len, max, fit — the reference's c<i>_fit_data.rds, :422-439) are appended to ``fit_data`` when given."""
    if nclust != len(sr_links):
        raise ValueError("Cluster mismatch detected, stopping!")
    main, dups = [], []
    for ci in range(1, nclust + 1):
        t = sr_links[ci - 1]
        t = t[t["len"].notna() & (t["len"] < sr_dist) & (t["len"] > 0)]
        # per-length 95th percentile (type 7 == pandas' linear interpolation), sorted by len   (:422)
        maxvls = t.groupby("len", sort=True)["MI"].quantile(0.95)
        mean_dist = fit_decay(maxvls.index.to_numpy(dtype=np.float64), maxvls.to_numpy())   # fastLm (:428-429)
        if fit_data is not None:
            fit_data.append(pd.DataFrame({"len": maxvls.index.to_numpy(dtype=np.float64), "max": maxvls.to_numpy(), "fit": mean_dist}))
        li = t["len"].to_numpy().astype(np.int64)                                  # positional index (Q5)
        ok = (li >= 1) & (li <= len(mean_dist))
        md = np.full(len(li), np.nan)
        md[ok] = mean_dist[li[ok] - 1]
        diff = t["MI"].to_numpy() - md
        idx = np.nonzero(diff > 0)[0]
        a_, b_ = beta_mle(diff[idx])                                               # (:452)
        t = t.iloc[idx].copy()
        t["srp_max"] = neg_log_beta_sf(diff[idx], a_, b_)                          # (:453, natural log: Q6)
        t.insert(0, "clust_c", ci)
        isdup = (t["clust1"] != t["clust2"]).to_numpy()
        main.append(t[~isdup])
        dups.append(t[isdup])
    df = pd.concat(main, ignore_index=True)
    dup = pd.concat(dups, ignore_index=True)
    if len(dup):
        # .I[which.max(srp_max)] by (pos1,pos2,clust1,clust2,len,MI), groups in order of first appearance (:478-485)
        first_max = dup.groupby(COLS, sort=False)["srp_max"].idxmax()
        df = pd.concat([df, dup.loc[first_max.to_numpy()]], ignore_index=True)
    red = df[df["srp_max"] > srp_cutoff]
    chk = df[df["MI"] >= red["MI"].min()] if len(red) else df.iloc[:0]
    return red.reset_index(drop=True), chk.reset_index(drop=True)
