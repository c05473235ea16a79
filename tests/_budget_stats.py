"""Statistics that tie a budgeted k-hop sampler to the reference's draw law (test infrastructure).

The reference draws `budget` times `rand() % deg` per frontier node whose degree exceeds the budget -- uniform over the row,
WITH replacement, the level's draws united into a set (ParallelSampler.cpp:528-540).  glibc's rand() stream cannot be
reproduced under this build's Philox contract, so budgeted calls are tied to the reference by what the
law implies.  `tests/golden/sampler_budget_hub_stats.npz` (oracle/gen_golden.py::gen_budget_hub_stats) holds the reference's
own outcome over 256 one-thread runs on a graph whose hubs have degree >> budget; this module turns a sampler's runs on
the same graph / roots into the same tables and compares

  * depth 1 against EXACT theory (occupancy law of 20 balls in deg bins): the number of distinct neighbours drawn (a
    without-replacement pick would always give 20), and the inclusion frequency by position in the row (a biased pick
    -- low offsets preferred, a modulo of a short word, an off-by-one at the row end -- tilts the deciles);
  * depth 1 and depth 2 against the reference's tables: per (root, node) inclusion counts (homogeneity chi-square) and
    the per-root distribution of subgraph sizes (two-sample Kolmogorov-Smirnov).

Every sampler here is seeded, so the statistics are fixed numbers: the bounds are several standard deviations wide and a
pass is a pass forever.
"""
import os

import numpy as np

from tests._golden import GOLDEN


def load():
    z = np.load(os.path.join(GOLDEN, "sampler_budget_hub_stats.npz"))
    return {k: z[k] for k in z.files}


def tables(sample_call, fx, depth):
    """counts[root, node], sizes[rep, root] of `reps` runs; ``sample_call(rep, depth)`` returns the per-subgraph node lists
    of one call over fx['roots'] (any sequence of 1-D integer arrays)."""
    reps, R, N = int(fx["reps"]), fx["roots"].size, fx["indptr"].size - 1
    counts = np.zeros((R, N), dtype=np.int64)
    sizes = np.zeros((reps, R), dtype=np.int64)
    for rep in range(reps):
        for r, nodes in enumerate(sample_call(rep, depth)):
            nodes = np.asarray(nodes).astype(np.int64)
            assert np.unique(nodes).size == nodes.size
            counts[r, nodes] += 1
            sizes[rep, r] = nodes.size
    return counts, sizes


def occupancy_moments(d, b):
    """Mean and variance of the number of distinct bins hit by b uniform balls in d bins."""
    q1, q2 = (1.0 - 1.0 / d) ** b, (1.0 - 2.0 / d) ** b
    mean = d * (1.0 - q1)
    var = d * q1 + d * (d - 1.0) * q2 - (d * q1) ** 2
    return mean, var


def depth1_theory(counts, sizes, fx):
    """Against the exact law, for the roots whose degree exceeds the budget: (a) z-scores of the mean number of distinct
    neighbours per root and their chi-square sum, the pooled variance ratio; (b) the inclusion counts by decile of the position in the row,
    chi-square against uniform; (c) roots with degree <= budget take every neighbour every time."""
    ip, ix, roots, b, reps = fx["indptr"].astype(np.int64), fx["indices"].astype(np.int64), fx["roots"].astype(np.int64), int(fx["budget"]), int(fx["reps"])
    z2, var_ratio, dec_obs, dec_exp, dec_var = [], [], np.zeros(10), np.zeros(10), np.zeros(10)
    full_ok = True
    for r, v in enumerate(roots):
        nb = ix[ip[v]:ip[v + 1]]
        d = nb.size
        assert v not in nb                                   # (no self loops: size - 1 = distinct neighbours)
        k = sizes[:, r] - 1
        if d <= b:
            full_ok &= bool(np.all(k == d) and np.all(counts[r, nb] == reps))
            continue
        mean, var = occupancy_moments(d, b)
        z2.append((k.mean() - mean) ** 2 / (var / reps))
        var_ratio.append(k.var(ddof=1) / var)
        p = 1.0 - (1.0 - 1.0 / d) ** b
        dec = np.minimum((np.arange(d) * 10) // d, 9)
        np.add.at(dec_obs, dec, counts[r, nb])
        np.add.at(dec_exp, dec, reps * p)
        # variance of a decile's count: indicators within one run are negatively correlated (occupancy law)
        q1, q2 = (1.0 - 1.0 / d) ** b, (1.0 - 2.0 / d) ** b
        m = np.bincount(dec, minlength=10).astype(np.float64)
        np.add.at(dec_var, np.arange(10), reps * (m * q1 * (1 - q1) + m * (m - 1) * (q2 - q1 * q1)))
        assert counts[r].sum() - reps == counts[r, nb].sum()  # nothing outside the row (+ the root itself, every run)
    z2 = np.asarray(z2)
    return dict(n_roots=int(z2.size), mean_chi2=float(z2.sum()), mean_chi2_dof=int(z2.size), max_abs_z=float(np.sqrt(z2.max())),
                var_ratio=float(np.mean(var_ratio)),
                decile_chi2=float((((dec_obs - dec_exp) ** 2) / dec_var).sum()), decile_dof=10, full_rows_ok=full_ok)


def homogeneity(c_ref, c_mine, min_total=16):
    """Sum over the (root, node) cells with at least `min_total` inclusions in the two tables together of
    (a - b)^2 / (a + b) * reps / (reps - (a + b) / 2)  -- two binomial(reps, p) counts with a common p: mean ~ 1 per cell."""
    a, b = c_ref.astype(np.float64), c_mine.astype(np.float64)
    reps = float(max(a.max(), b.max()))
    tot = a + b
    sel = (tot >= min_total) & (tot <= 2 * reps - min_total)         # (cells that are always / never in: no information)
    stat = ((a - b)[sel] ** 2 / tot[sel]) * (reps / (reps - tot[sel] / 2.0))
    return float(stat.sum()), int(sel.sum())


def ks_two_sample(x, y):
    """Two-sample Kolmogorov-Smirnov statistic scaled by sqrt(n m / (n + m)) (asymptotic law: P(K > 1.95) ~ 1e-3)."""
    x, y = np.sort(np.asarray(x, dtype=np.float64)), np.sort(np.asarray(y, dtype=np.float64))
    grid = np.concatenate([x, y])
    fx_ = np.searchsorted(x, grid, side="right") / x.size
    fy_ = np.searchsorted(y, grid, side="right") / y.size
    return float(np.abs(fx_ - fy_).max() * np.sqrt(x.size * y.size / (x.size + y.size)))


def against_reference(counts, sizes, fx, depth):
    ref_c, ref_s = fx[f"d{depth}_counts"], fx[f"d{depth}_sizes"]
    stat, dof = homogeneity(ref_c, counts)
    ks = [ks_two_sample(ref_s[:, r], sizes[:, r]) for r in range(sizes.shape[1]) if ref_s[:, r].min() != ref_s[:, r].max()]
    return dict(homog=stat, homog_dof=dof, homog_ratio=stat / max(dof, 1), ks_max=float(max(ks)), ks_roots=len(ks),
                mean_size_ref=float(ref_s.mean()), mean_size=float(sizes.mean()))


def check(report_theory, report_d1, report_d2):
    """The bounds (fixed seeds: deterministic numbers; widths in standard deviations of the statistic under the law)."""
    t = report_theory
    assert t["full_rows_ok"], "a root of degree <= budget did not take all its neighbours"
    # sum of n z^2 ~ chi2(n): mean n, sd sqrt(2 n); 48 roots -> 48 +- 9.8; a without-replacement pick gives ~1e4
    assert t["mean_chi2"] < t["mean_chi2_dof"] + 5 * np.sqrt(2 * t["mean_chi2_dof"]), t
    assert t["max_abs_z"] < 4.5, t
    assert 0.85 < t["var_ratio"] < 1.15, t
    assert t["decile_chi2"] < 10 + 5 * np.sqrt(20), t        # chi2(10) (9 if the total were conditioned on): 10 +- 4.5
    for rep in (report_d1, report_d2):
        sd = np.sqrt(2.0 / rep["homog_dof"])
        assert rep["homog_dof"] > 1000 and abs(rep["homog_ratio"] - 1.0) < 6 * sd + 0.02, rep
        assert rep["ks_max"] < 2.2, rep                      # (P(K > 2.2) ~ 1.2e-4 per root; 48-64 roots)
        assert abs(rep["mean_size"] - rep["mean_size_ref"]) / rep["mean_size_ref"] < 0.01, rep
