"""The oracle against the committed golden vectors, scikit-learn and scipy.
(CPU only; the oracle is test infrastructure, never the product path.)"""
import numpy as np
import pytest

import lc_oracle as o

WF = {"Dirichlet": o.Dirichlet, "StickBreak": o.StickBreak, "GDirichlet": o.GDirichlet}


def test_constants_are_float_literals():
    # include/libcluster.h:125-127 -- 1e-5f / 0.1f widened to double
    assert o.CONVERGE == 9.999999747378752e-06
    assert o.ZEROCUTOFF == 0.10000000149011612
    assert o.FENGYDEL == o.CONVERGE / 10


def test_oracle_reproduces_golden_cases(estep_cases):
    for c in estep_cases:
        X = [np.array(x) for x in c["X"]]
        q0 = [np.array(q) for q in c["q0"]]
        Ftr, Fztr, qT, w, cl = o.vbem_fixed(X, q0, WF[c["weights"]], c["prior"], c["iters"], c["sparse"])
        np.testing.assert_allclose(Ftr, c["Ftrace"], rtol=1e-12)
        np.testing.assert_allclose(Fztr, c["Fztrace"], rtol=1e-12)
        for a, b in zip(qT, c["qT"]):
            np.testing.assert_allclose(a, np.array(b), rtol=1e-10, atol=1e-300)


def test_suffstats_equal_dense_products(estep_cases):
    for c in estep_cases:
        if c["sparse"]:
            continue
        X = np.vstack([np.array(x) for x in c["X"]])
        q = np.vstack([np.array(x) for x in c["q0"]])
        Nk, xs, xxs = o.suffstats(X, q)
        np.testing.assert_allclose(Nk, c["stats"]["Nk"], rtol=1e-12)
        np.testing.assert_allclose(xs, np.array(c["stats"]["xs"]), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(xxs, np.array(c["stats"]["xxs"]), rtol=1e-10, atol=1e-10)


def test_qz_rows_sum_to_one_and_fz_is_logsumexp(estep_cases):
    for c in estep_cases:
        for q in c["q1"]:
            q = np.array(q)
            if q.size:
                np.testing.assert_allclose(q.sum(axis=1), 1.0, rtol=1e-12)
        if not c["sparse"]:
            tot = 0.0
            for j in range(c["J"]):
                lq = np.array(c["Elogpi"][j])[None, :] + np.array(c["Eloglike"][j])
                tot -= o.logsumexp(lq).sum()
            assert abs(tot - c["Fz1"]) <= 1e-10 * abs(c["Fz1"])


def test_eloglike_matches_sklearn():
    """GaussWish::Eloglike + Dirichlet::Elogweight == scikit-learn's
    BayesianGaussianMixture E-step (Bishop 10.2), an independent implementation."""
    from sklearn.mixture import BayesianGaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky

    rng = np.random.default_rng(7)
    N, D, K = 400, 8, 4
    X = rng.normal(size=(N, D)) + rng.integers(0, 3, (N, 1)) * 2.0
    q = rng.dirichlet(np.ones(K), N)
    w = o.Dirichlet()
    cl = [o.GaussWish(1.0, D) for _ in range(K)]
    w.update(o.updateSS(X, q, cl))
    for c in cl:
        c.update()
    ours = np.stack([c.Eloglike(X) for c in cl], axis=1)
    bgm = BayesianGaussianMixture(n_components=K, covariance_type="full",
                                  weight_concentration_prior_type="dirichlet_distribution")
    bgm.mean_precision_ = np.array([c.beta for c in cl])
    bgm.means_ = np.stack([c.m for c in cl])
    bgm.degrees_of_freedom_ = np.array([c.nu for c in cl])
    bgm.covariances_ = np.stack([c.iW / c.nu for c in cl])
    bgm.precisions_cholesky_ = _compute_precision_cholesky(bgm.covariances_, "full")
    bgm.weight_concentration_ = w.alpha
    np.testing.assert_allclose(ours, bgm._estimate_log_prob(X), rtol=0, atol=1e-10)
    np.testing.assert_allclose(w.Elogweight(), bgm._estimate_log_weights(), rtol=0, atol=1e-13)


def test_stickbreak_matches_sklearn_when_sorted():
    """sklearn's dirichlet_process weights walk components in index order;
    StickBreak walks them by descending size (distributions.cpp:141-167): equal when pre-sorted."""
    from scipy.special import digamma

    Nk = np.array([50.0, 30.0, 15.0, 5.0])
    sb = o.StickBreak(1.0)
    sb.update(Nk)
    a, b = 1.0 + Nk, 1.0 + np.hstack((np.cumsum(Nk[::-1])[-2::-1], 0))
    dsum = digamma(a + b)
    ref = digamma(a) - dsum + np.hstack((0, np.cumsum(digamma(b) - dsum)[:-1]))
    np.testing.assert_allclose(sb.Elogweight(), ref, rtol=1e-13)


def test_learners_on_reference_test_data(xcat, xcat_traces):
    """test/testdata.h data; expectations = SURVEY Appendix B (restatement-derived)."""
    F, qZ, w, cl = o.learnBGMM(xcat["Xcat"])
    t = xcat_traces["learnBGMM"]
    assert len(cl) == t["K"] == 3
    assert abs(F - 567.973353292) < 1e-6 and abs(F - t["F"]) < 1e-9
    F, qZ, w, cl = o.learnVDP(xcat["Xcat"])
    assert len(cl) == 3 and abs(F - 571.970406140) < 1e-6
    F, qZ, w, cl = o.learnGMC(xcat["X"])
    assert len(cl) == 4 and abs(F - 534.952781820) < 1e-6
    F, _, _, cl = o.learnBGMM(xcat["Xcat"], maxclusters=1)  # the README's accidental call (README.md:205)
    assert len(cl) == 1 and abs(F - 800.200434814) < 1e-6


def test_maxit_off_by_one(xcat):
    # cluster.cpp:235-236: maxit=1 runs two iterations when not converged
    X = [xcat["Xcat"]]
    rng = np.random.default_rng(0)
    q = rng.dirichlet(np.ones(3), 120)
    tr = []
    o.vbem(X, [q], [], [], 1.0, maxit=1, wfactory=o.Dirichlet, trace=tr)
    assert len(tr) == 2


def test_errors():
    with pytest.raises(ValueError):
        o.GaussWish(0.0, 2)
    with pytest.raises(ValueError):
        o.Dirichlet(0.0)
    with pytest.raises(ValueError):
        o.StickBreak(-1.0)


def test_std_sort_restatement_matches_libstdcxx(tmp_path):
    """The oracle restates libstdc++'s std::sort (unstable; the reference sorts the stick-breaking order and the
    split candidates with it).  Compiled here against the real thing on tie-heavy inputs of 1 ... 300 elements."""
    import random
    import subprocess

    src = tmp_path / "s.cpp"
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <utility>
#include <vector>
int main() {
  int n;
  while (scanf("%d", &n) == 1) {
    std::vector<std::pair<int, double> > v(n);
    for (int i = 0; i < n; ++i) { v[i].first = i; if (scanf("%lf", &v[i].second) != 1) return 1; }
    std::sort(v.begin(), v.end(),
              [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.second > b.second; });
    for (int i = 0; i < n; ++i) printf("%d ", v[i].first);
    printf("\n");
  }
  return 0;
}
''')
    exe = tmp_path / "s"
    subprocess.run(["g++", "-O2", "-o", str(exe), str(src)], check=True)
    rnd = random.Random(3)
    cases = []
    for _ in range(300):
        n = rnd.choice([1, 2, 5, 16, 17, 18, 31, 33, 40, 65, 100, 300])
        cases.append([float(rnd.randint(0, max(1, n // 3))) for _ in range(n)])
    inp = "\n".join(f"{len(v)} " + " ".join(map(str, v)) for v in cases) + "\n"
    out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    for vals, line in zip(cases, out):
        assert o.std_sort(range(len(vals)), lambda i, j: vals[i] > vals[j]) == [int(x) for x in line.split()]
