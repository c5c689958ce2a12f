"""The reference's Python interface shape (python/libclusterpy.h:300-470, exercised by python/testapi.py):
keyword names, defaults and the exact return tuples of the Boost.Python module, on the HIP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MEANS = np.array([[0.0, 0.0], [5.0, 5.0], [-5.0, -5.0]])
BETA = np.array([[1 / 3, 1 / 3, 1 / 3], [1 / 2, 1 / 4, 1 / 4], [1 / 4, 1 / 4, 1 / 2]])


def _gmm(rng, n, weights=None):
    pi = rng.random(3) + 0.2 if weights is None else np.asarray(weights, dtype=float)
    pi = pi / pi.sum()
    nk = np.round(pi * n).astype(int)
    nk[-1] = n - nk[:-1].sum()
    return np.concatenate([rng.multivariate_normal(MEANS[k], np.eye(2), int(nk[k])) for k in range(3)])


def _check_gmm(w, mu, cov, K=3):
    assert len(mu) == K and len(cov) == K
    assert all(m.shape == (1, 2) for m in mu) and all(c.shape == (2, 2) for c in cov)  # RowVectorXd / MatrixXd
    got = np.vstack(mu)
    for m in MEANS:
        assert np.min(np.linalg.norm(got - m, axis=1)) < 0.3
    for c in cov:
        np.testing.assert_allclose(c, np.eye(2), atol=0.35)


def test_flat_and_grouped_mixtures_like_testapi():
    import libcluster_amd as lc

    rng = np.random.default_rng(11)
    W = _gmm(rng, 10000)
    f, qZ, w, mu, cov = lc.learnVDP(W, verbose=False)                      # 5-tuple, libclusterpy.cpp:157-158
    assert qZ.shape == (10000, 3) and w.shape == (3, 1) and np.isfinite(f)
    _check_gmm(w, mu, cov)
    f, qZ, w, mu, cov = lc.learnBGMM(W, prior=1.0, maxclusters=-1, verbose=False, threads=4)
    assert abs(w.sum() - 1.0) < 1e-3
    _check_gmm(w, mu, cov)

    Wg = [_gmm(rng, 2000) for _ in range(4)]
    f, qZ, w, mu, cov = lc.learnGMC(Wg, sparse=False, threads=2)
    assert isinstance(qZ, list) and len(qZ) == 4 and qZ[0].shape == (2000, 3)
    assert isinstance(w, list) and w[0].shape == (3, 1)
    _check_gmm(w, mu, cov)
    f, qZ, w, mu, cov = lc.learnSGMC(Wg)
    _check_gmm(w, mu, cov)


def test_multiple_level_models_like_testapi():
    import libcluster_amd as lc

    rng = np.random.default_rng(12)
    I, Ni = 200, 100
    Y = rng.integers(0, 3, I)
    W = np.stack([rng.multivariate_normal(MEANS[Y[i]], np.eye(2)) for i in range(I)])
    X = [_gmm(rng, Ni, BETA[Y[i]]) for i in range(I)]
    f, qY, qZ, wi, ws, mu, cov = lc.learnSCM([X], trunc=30, verbose=False)  # 7-tuple, libclusterpy.cpp:270-271
    assert len(qY) == 1 and qY[0].shape[0] == I and len(qZ) == 1 and len(qZ[0]) == I
    T = qY[0].shape[1]
    assert 1 <= T <= 30 and len(ws) == T and wi[0].shape == (T, 1) and ws[0].shape == (len(mu), 1)
    _check_gmm(ws, mu, cov)
    f, qY, qZ, wi, ws, mui, mus, covi, covs = lc.learnMCM([W], [X], trunc=30)  # 9-tuple, libclusterpy.cpp:305-307
    assert len(mui) == len(covi) == qY[0].shape[1] and mui[0].shape == (1, 2)
    # Without qY0 the start is the reference's: Eigen's Random() = std::rand() (scluster.cpp:519-521), whose state depends
    # on everything this process has drawn before (RCCL's bootstrap draws too) -- shapes only.  The clusters are checked
    # on a reproducible start of the same form (|U(-1,1)| rows, normalised).
    r = np.abs(np.random.default_rng(0).uniform(-1.0, 1.0, (I, 30)))
    f, qY, qZ, wi, ws, mui, mus, covi, covs = lc.learnMCM([W], [X], trunc=30, qY0=[r / r.sum(axis=1, keepdims=True)])
    assert len(mui) == len(covi) == qY[0].shape[1] == 3
    _check_gmm(ws, mus, covs)
    # the document-level Gaussians sit on the three document classes
    got = np.vstack(mui)
    assert all(np.min(np.linalg.norm(MEANS - g, axis=1)) < 1.0 for g in got)


def test_prior_passes_through_single_precision_like_the_binding():
    """`const float clusterprior` (libclusterpy.h:74-135): 0.37 becomes float(0.37f) before the learner sees it."""
    import libcluster_amd as lc
    from libcluster_amd import capi

    rng = np.random.default_rng(13)
    W = _gmm(rng, 3000)
    f1, *_ = lc.learnBGMM(W, prior=0.37)
    F2, m, _ = capi.learn(capi.ALGO_BGMM, W, 1.0, float(np.float32(0.37)), -1, False, False, 2, 0)
    m.close()
    assert f1 == F2
