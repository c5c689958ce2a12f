"""learnSCM / learnMCM (src/scluster.cpp, src/mcluster.cpp) on the HIP path: every document is one group of the
device context, vbeZ is the E-step kernel with a per-document constant table.  Checked against the oracle and the
committed traces of the reference's own test set-ups (test/scluster_test.cpp, test/mcluster_test.cpp)."""
import json

import numpy as np
import pytest

import lc_oracle as o
from conftest import GOLDEN
from test_gpu_parity import assert_q_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def top():
    return json.loads((GOLDEN / "topic_traces.json").read_text())


def _check_rounds(got, ref):
    assert [(t, k) for t, k, _ in got] == [(t, k) for t, k, _ in ref]
    for (_, _, a), (_, _, b) in zip(got, ref):
        np.testing.assert_allclose(a, b, rtol=1e-8)


def _check_q(qY, qZ, rqY, rqZ):
    for a, b in zip(qY, rqY):
        assert_q_close(a, np.array(b), rtol=1e-6)
    for ga, gb in zip(qZ, rqZ):
        assert len(ga) == len(gb)
        for a, b in zip(ga, gb):
            assert_q_close(a, np.array(b), rtol=1e-6)


def test_learnSCM_on_reference_test_setup(xcat, top):
    """test/scluster_test.cpp:44-68: 2 groups x 6 documents of testdata.h, maxT = 4."""
    import libcluster_amd as lc

    ref = top["learnSCM"]
    X = xcat["X"]
    F, qY, qZ, wj, wt, means, covs, info = lc.learnSCM([X[:6], X[6:]], trunc=ref["maxT"], qY0=ref["qY0"], return_info=True)
    assert (info["T"], info["K"]) == (ref["T"], ref["K"])
    assert abs(F - ref["F"]) <= 1e-8 * abs(ref["F"])
    _check_rounds(info["rounds"], ref["rounds"])
    _check_q(qY, qZ, ref["qY"], ref["qZ"])
    np.testing.assert_allclose(np.vstack(means), np.array(ref["means_k"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(covs), np.array(ref["covs_k"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(info["Elogweight_j"]), np.array(ref["Elogweight_j"]), rtol=1e-8)
    np.testing.assert_allclose(np.array(info["Elogweight_t"]), np.array(ref["Elogweight_t"]), rtol=1e-8)
    np.testing.assert_allclose(np.hstack(wj).T, np.exp(np.array(ref["Elogweight_j"])), rtol=1e-8)


def test_learnMCM_on_reference_test_setup(xcat, top):
    """test/mcluster_test.cpp:44-70: the same documents plus the O data as document observations, maxT = 10."""
    import libcluster_amd as lc

    ref = top["learnMCM"]
    X = xcat["X"]
    F, qY, qZ, wj, wt, mt, mk, ct, ck, info = lc.learnMCM(xcat["O"], [X[:6], X[6:]], trunc=ref["maxT"],
                                                         qY0=ref["qY0"], return_info=True)
    assert (info["T"], info["K"]) == (ref["T"], ref["K"])
    assert abs(F - ref["F"]) <= 1e-8 * abs(ref["F"])
    _check_rounds(info["rounds"], ref["rounds"])
    _check_q(qY, qZ, ref["qY"], ref["qZ"])
    np.testing.assert_allclose(np.vstack(mk), np.array(ref["means_k"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(ck), np.array(ref["covs_k"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.vstack(mt), np.array(ref["means_t"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(ct), np.array(ref["covs_t"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(info["Elogweight_t"]), np.array(ref["Elogweight_t"]), rtol=1e-8)


def _synthetic(rng, J, I, n, D, K, T, Dt=0):
    """Documents drawn from T document classes, each a different mixture over K Gaussian segments."""
    mu = rng.normal(0, 5.0, (K, D))
    mix = rng.dirichlet(np.full(K, 0.4), T)
    mw = rng.normal(0, 4.0, (T, max(Dt, 1)))
    X, W = [], []
    for j in range(J):
        Xj, Wj = [], []
        for i in range(I):
            t = rng.integers(0, T)
            nn = n + int(rng.integers(-n // 3, n // 3 + 1))
            z = rng.choice(K, size=nn, p=mix[t])
            Xj.append(mu[z] + rng.normal(size=(nn, D)))
            Wj.append(mw[t] + 0.7 * rng.normal(size=max(Dt, 1)))
        X.append(Xj)
        W.append(np.array(Wj))
    return X, (W if Dt else None)


@pytest.mark.parametrize("mcm", [False, True])
def test_topic_models_match_oracle_on_synthetic_documents(mcm):
    import libcluster_amd as lc

    rng = np.random.default_rng(17 + mcm)
    J, I, maxT = 3, 7, 5
    X, W = _synthetic(rng, J, I, 90, 3, 4, 2, Dt=2 if mcm else 0)
    X[1][2] = X[1][2][:1]  # a one-row document (ragged, sub-16 group)
    qY0 = [o.random_qY(I, maxT, rng) for _ in range(J)]
    tr = []
    if mcm:
        Fo, qYo, qZo, wjo, wto, cto, clo = o.learnMCM(W, X, maxT=maxT, qY0=qY0, trace=tr)
        F, qY, qZ, wj, wt, mt, mk, ct, ck, info = lc.learnMCM(W, X, trunc=maxT, qY0=qY0, nthreads=2, return_info=True)
        np.testing.assert_allclose(np.vstack(mt), np.array([c.getmean() for c in cto]), rtol=1e-7, atol=1e-9)
    else:
        Fo, qYo, qZo, wjo, wto, clo = o.learnSCM(X, maxT=maxT, qY0=qY0, trace=tr)
        F, qY, qZ, wj, wt, mk, ck, info = lc.learnSCM(X, trunc=maxT, qY0=qY0, nthreads=2, return_info=True)
    assert (info["T"], info["K"]) == (len(wto), len(clo))
    _check_rounds(info["rounds"], tr)
    assert abs(F - Fo) <= 1e-8 * abs(Fo)
    _check_q(qY, qZ, qYo, qZo)
    np.testing.assert_allclose(np.vstack(mk), np.array([c.getmean() for c in clo]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(ck), np.array([c.getcov() for c in clo]), rtol=1e-7, atol=1e-8)


def test_topic_argument_errors_and_random_start(xcat):
    import libcluster_amd as lc

    X = xcat["X"]
    Xv = [X[:6], X[6:]]
    with pytest.raises(ValueError, match="maxT must be less than the number of documents"):
        lc.learnSCM(Xv, trunc=13)  # scluster.cpp:531-533 (12 documents)
    with pytest.raises(ValueError, match="same number of groups"):
        lc.learnMCM(xcat["O"][:1], Xv, trunc=4)  # mcluster.cpp:548-549
    with pytest.raises(ValueError, match="same number of 'docs'"):
        lc.learnMCM([xcat["O"][0], xcat["O"][1][:5]], Xv, trunc=4)  # mcluster.cpp:553-555
    with pytest.raises(ValueError, match="at least one thread"):
        lc.learnSCM(Xv, trunc=4, nthreads=0)
    # the reference's own start (std::rand): a valid model comes back
    F, qY, qZ, wj, wt, means, covs = lc.learnSCM(Xv, trunc=4)  # the reference tuple (libclusterpy.cpp:270)
    info = {"T": qY[0].shape[1], "K": len(means)}
    assert np.isfinite(F) and 1 <= info["T"] <= 4 and info["K"] >= 1
    for q in qY:
        np.testing.assert_allclose(q.sum(axis=1), 1.0, rtol=1e-9)
    for g in qZ:
        for q in g:
            np.testing.assert_allclose(q.sum(axis=1), 1.0, rtol=1e-9)
