"""Short-range p-value model and link merge: host logic of ``mergeNsort_sr_links``
(R/computePairwiseMI.R:400-495).  Runs on the host like the reference (it is O(#sr links) table work plus
a two-parameter optimiser); the device-resident version is listed as "next" in DESIGN.md.

Reproduced quirks: Q5 ``mean_dist[len]`` is indexed by the VALUE of ``len`` (:448) and Q6 ``srp_max`` is
a natural-log tail probability (:453).  The plot / RDS side outputs (:430-440) are not produced.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

COLS = ["pos1", "pos2", "clust1", "clust2", "len", "MI"]


def beta_mle(x: np.ndarray):
    """coef(fitdistrplus::fitdist(x, "beta")): moment start + Nelder-Mead on the log-likelihood (optim default)."""
    from scipy import optimize, special
    n = x.size
    if n < 2:  # fitdistrplus::fitdist stops the same way; happens when the fit of (:428) is NaN or no link exceeds it
        raise ValueError("fitdist: data must be a numeric vector of length greater than 1 (no short-range link exceeds the fitted decay)")
    m = float(np.mean(x))
    v = (n - 1) / n * float(np.var(x, ddof=1))
    aux = m * (1 - m) / v - 1
    slx, sl1x = float(np.sum(np.log(x))), float(np.sum(np.log1p(-x)))

    def nll(p):
        a, b = p
        if a <= 0 or b <= 0:
            return np.inf
        return -((a - 1) * slx + (b - 1) * sl1x - n * special.betaln(a, b))

    res = optimize.minimize(nll, np.array([m * aux, (1 - m) * aux]), method="Nelder-Mead",
                            options=dict(xatol=1e-10, fatol=1e-12, maxiter=5000, maxfev=10000))
    # optim() stops at reltol 1e-8 on the log-likelihood; polish to the stationary point with Newton steps on the
    # score equations so that the result does not depend on the simplex path (deterministic to ~1e-12)
    a, b = float(res.x[0]), float(res.x[1])
    for _ in range(50):
        ga = n * (special.digamma(a + b) - special.digamma(a)) + slx
        gb = n * (special.digamma(a + b) - special.digamma(b)) + sl1x
        tab = special.polygamma(1, a + b)
        haa, hbb, hab = n * (tab - special.polygamma(1, a)), n * (tab - special.polygamma(1, b)), n * tab
        det = haa * hbb - hab * hab
        da, db = (hbb * ga - hab * gb) / det, (haa * gb - hab * ga) / det
        if not (np.isfinite(da) and np.isfinite(db)) or a - da <= 0 or b - db <= 0:
            break
        a, b = a - da, b - db
        if abs(da) < 1e-13 * a and abs(db) < 1e-13 * b:
            break
    return a, b


def merge_n_sort_sr_links(sr_links: list, nclust: int, sr_dist: float, srp_cutoff: float):
    """sr_links: list (one per cluster) of DataFrames with COLS.  Returns (sr_links_red, sr_links_ARACNE_check)."""
    from scipy import stats
    if nclust != len(sr_links):
        raise ValueError("Cluster mismatch detected, stopping!")
    main, dups = [], []
    for ci in range(1, nclust + 1):
        t = sr_links[ci - 1]
        t = t[t["len"].notna() & (t["len"] < sr_dist) & (t["len"] > 0)]
        # per-length 95th percentile (type 7 == pandas' linear interpolation), sorted by len   (:422)
        maxvls = t.groupby("len", sort=True)["MI"].quantile(0.95)
        ulen = maxvls.index.to_numpy(dtype=np.float64)
        X = np.column_stack([np.log(ulen), np.ones(len(ulen))])
        coef, *_ = np.linalg.lstsq(X, np.log(maxvls.to_numpy()), rcond=None)     # fastLm (:428)
        mean_dist = np.exp(X @ coef)                                               # (:429)
        li = t["len"].to_numpy().astype(np.int64)                                  # positional index (Q5)
        ok = (li >= 1) & (li <= len(mean_dist))
        md = np.full(len(li), np.nan)
        md[ok] = mean_dist[li[ok] - 1]
        diff = t["MI"].to_numpy() - md
        idx = np.nonzero(diff > 0)[0]
        a_, b_ = beta_mle(diff[idx])                                               # (:452)
        t = t.iloc[idx].copy()
        t["srp_max"] = -stats.beta.logsf(diff[idx], a_, b_)                        # (:453, natural log: Q6)
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
