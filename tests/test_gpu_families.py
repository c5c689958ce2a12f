"""Diagonal-Gaussian (NormGamma) and exponential (ExpGamma) cluster families on the HIP path, against the oracle
and the committed traces (tests/golden/family_traces.json).  SURVEY 8(f) rank 3: learnDGMM / learnBEMM /
learnDGMC / learnEGMC (src/cluster.cpp:697-873, src/distributions.cpp:418-590)."""
import json

import numpy as np
import pytest

import lc_oracle as o
from libcluster_amd import capi
from conftest import GOLDEN
from test_gpu_parity import RTOL_F, assert_q_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fam():
    return json.loads((GOLDEN / "family_traces.json").read_text())


def _data(rng, N, D, K, J, positive):
    X, q0 = [], []
    for j in range(J):
        n = N // J + (j == 0) * (N % J)
        if positive:
            X.append(rng.exponential(1.0, (n, D)) * rng.uniform(0.2, 5.0, (1, D)))
        else:
            X.append(rng.normal(size=(n, D)) * 1.5 + rng.integers(0, K, (n, 1)))
        q0.append(rng.dirichlet(np.ones(K) * 0.3, n))
    return X, q0


def _diag_params(cl):
    """(a, w2, w1, const) of lc_estep_diag for oracle clusters (include/libcluster_hip.h)."""
    from scipy.special import digamma

    K, D = len(cl), cl[0].D
    a, w2, w1, cst = np.zeros((K, D)), np.zeros((K, D)), np.zeros((K, D)), np.zeros(K)
    for k, c in enumerate(cl):
        if isinstance(c, o.NormGamma):
            a[k], w2[k] = c.m, -0.5 * c.nu / c.L
            cst[k] = 0.5 * (D * (digamma(c.nu) - np.log(2 * np.pi) - 1.0 / c.beta) - c.logL)
        else:
            w1[k] = -c.a * c.ib
            cst[k] = D * digamma(c.a) - c.logb
    return a, w2, w1, cst


@pytest.mark.parametrize("family", ["NormGamma", "ExpGamma"])
@pytest.mark.parametrize("N,D,K,J", [(1000, 16, 8, 1), (777, 23, 5, 3), (513, 64, 6, 1), (300, 128, 3, 2),
                                      (4099, 2, 2, 1), (50, 7, 33, 1), (2500, 1, 4, 2), (1300, 100, 40, 1)])
def test_diag_suffstat_and_estep_vs_oracle(family, N, D, K, J):
    rng = np.random.default_rng(N + D + K)
    eg = family == "ExpGamma"
    X, q0 = _data(rng, N, D, K, J, eg)
    cf = o.ExpGamma if eg else o.NormGamma
    weights = [(o.GDirichlet if J > 1 else o.Dirichlet)() for _ in range(J)]
    cl = [cf(1.0, D) for _ in range(K)]
    for j in range(J):
        weights[j].update(o.updateSS(X[j], q0[j], cl))
    Nref, xsref = np.array([c.N_s for c in cl]), np.stack([c.x_s for c in cl])
    xxref = None if eg else np.stack([c.xx_s for c in cl])
    for c in cl:
        c.update()
    qref, Fzref, llref, Eref = [], 0.0, np.zeros(K), []
    for j in range(J):
        q, fz = o.vbexpectation(X[j], weights[j], cl)
        E = np.stack([c.Eloglike(X[j]) for c in cl], axis=1)
        qref.append(q)
        Eref.append(E)
        Fzref += fz
        llref += np.einsum("nk,nk->k", q, E)
    a, w2, w1, cst = _diag_params(cl)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        Nk, xs, xxs, Njk = ctx.suffstat_diag(second=not eg)
        np.testing.assert_allclose(Nk, Nref, rtol=1e-11)
        np.testing.assert_allclose(xs, xsref, rtol=1e-10, atol=1e-10)
        if not eg:
            np.testing.assert_allclose(xxs, xxref, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(Njk, np.stack([q.sum(axis=0) for q in q0]), rtol=1e-11)
        c = np.stack([w.Elogweight() for w in weights]) + cst[None, :]
        Fz, ll = ctx.estep_diag(a, w2, w1, c)
        q = ctx.get_qz([x.shape[0] for x in X])
        # raw mode = the Eloglike columns (no weights, no normalisation)
        ctx.estep_diag(a, w2, w1, np.tile(cst, (J, 1)), raw=True)
        E = ctx.get_qz([x.shape[0] for x in X])
    assert abs(Fz - Fzref) <= RTOL_F * abs(Fzref)
    for j in range(J):
        assert_q_close(q[j], qref[j])
        np.testing.assert_allclose(q[j].sum(axis=1), 1.0, rtol=1e-12)
        np.testing.assert_allclose(E[j], Eref[j], rtol=1e-10, atol=1e-9)
    # LLk excludes the per-cluster constant (lc_estep's convention): add it back
    got = ll + sum(qj.sum(axis=0) for qj in q) * cst
    np.testing.assert_allclose(got, llref, rtol=1e-9, atol=1e-8)


def test_diag_suffstat_sparse_mask():
    rng = np.random.default_rng(5)
    X, q0 = _data(rng, 900, 9, 5, 3, False)
    mask = np.ones((3, 5), dtype=np.uint8)
    mask[1, 2] = mask[2, 0] = 0
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        Nk, xs, xxs, Njk = ctx.suffstat_diag(mask)
    Nr, xr, xxr = np.zeros(5), np.zeros((5, 9)), np.zeros((5, 9))
    for j in range(3):
        for k in range(5):
            if mask[j, k]:
                Nr[k] += q0[j][:, k].sum()
                xr[k] += q0[j][:, k] @ X[j]
                xxr[k] += q0[j][:, k] @ (X[j] * X[j])
    np.testing.assert_allclose(Nk, Nr, rtol=1e-11)
    np.testing.assert_allclose(xs, xr, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(xxs, xxr, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(Njk, np.stack([q.sum(axis=0) for q in q0]), rtol=1e-11)  # Njk itself is unmasked


@pytest.mark.parametrize("family,wkind", [("NormGamma", capi.W_DIRICHLET), ("ExpGamma", capi.W_GDIRICHLET)])
def test_vbem_fixed_iterations_vs_oracle(family, wkind):
    rng = np.random.default_rng(31)
    eg = family == "ExpGamma"
    J = 1 if wkind == capi.W_DIRICHLET else 3
    X, q0 = _data(rng, 1500, 12, 6, J, eg)
    cf = o.ExpGamma if eg else o.NormGamma
    wf = o.Dirichlet if wkind == capi.W_DIRICHLET else o.GDirichlet
    tro, _, qo, wo, clo = o.vbem_fixed(X, q0, wf, 1.0, 4, False, cf)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        F, tr, model = ctx.vbem(wkind, 1.0, 1.0, fixed_iters=4, ckind=capi.C_EXPGAMMA if eg else capi.C_NORMGAMMA)
        q = ctx.get_qz([x.shape[0] for x in X])
        Ns = [model.cluster(k)["N"] for k in range(6)]
        Fw, Fc = model.fenergy()
        model.close()
    np.testing.assert_allclose(tr, tro, rtol=RTOL_F)
    for j in range(J):
        assert_q_close(q[j], qo[j], rtol=1e-8)
    np.testing.assert_allclose(Ns, [c.getN() for c in clo], rtol=1e-9)
    np.testing.assert_allclose(Fc, [c.fenergy() for c in clo], rtol=1e-9)
    np.testing.assert_allclose(Fw, [w.fenergy() for w in wo], rtol=1e-9)


def _check(res, ref):
    F, qZ, w, means, covs, info = res
    assert info["K"] == ref["K"]
    assert abs(F - ref["F"]) <= 1e-8 * abs(ref["F"])
    assert [k for k, _ in info["rounds"]] == [k for k, _ in ref["rounds"]]
    for (_, tr), (_, rtr) in zip(info["rounds"], ref["rounds"]):
        np.testing.assert_allclose(tr, rtr, rtol=1e-8)
    np.testing.assert_allclose(info["N"], ref["N"], rtol=1e-7)
    if "rates" in ref:
        np.testing.assert_allclose(np.vstack(means), np.array(ref["rates"]), rtol=1e-7)
        assert all(c is None for c in covs)
    else:
        np.testing.assert_allclose(np.vstack(means), np.array(ref["means"]), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(np.vstack(covs), np.array(ref["covs"]), rtol=1e-7, atol=1e-9)
    qs = qZ if isinstance(qZ, list) else [qZ]
    for a, b in zip(qs, ref["qZ"]):
        assert_q_close(a, np.array(b), rtol=1e-6)
    np.testing.assert_allclose(np.array(info["Elogweight"]), np.array(ref["Elogweight"]), rtol=1e-8)


def test_learnDGMM_and_DGMC_on_reference_test_data(xcat, fam):
    import libcluster_amd as lc

    _check(lc.learnDGMM(xcat["Xcat"], return_info=True), fam["learnDGMM"])
    _check(lc.learnDGMC(xcat["X"], return_info=True), fam["learnDGMC"])


def test_learnBEMM_and_EGMC(xcat, fam):
    import libcluster_amd as lc

    Xpos = [np.abs(g) + 0.1 for g in xcat["X"]]
    _check(lc.learnBEMM(np.vstack(Xpos), return_info=True), fam["learnBEMM"])
    _check(lc.learnEGMC(Xpos, return_info=True), fam["learnEGMC"])
    Xexp = np.array(fam["Xexp"])
    _check(lc.learnBEMM(Xexp, return_info=True), fam["learnBEMM_exp"])
    _check(lc.learnEGMC([Xexp[:130], Xexp[130:250], Xexp[250:]], return_info=True), fam["learnEGMC_exp"])


def test_exponential_learners_reject_negative_observations(xcat):
    """cluster.cpp:742-743, 862-864."""
    import libcluster_amd as lc

    with pytest.raises(ValueError, match=r"X has to be in the range \[0, inf\)!"):
        lc.learnBEMM(xcat["Xcat"])
    with pytest.raises(ValueError, match=r"X has to be in the range \[0, inf\)!"):
        lc.learnEGMC(xcat["X"])


@pytest.mark.parametrize("family", ["NormGamma", "ExpGamma"])
def test_cluster_on_device_resident_data(family):
    """lc_cluster with a cluster family on context-resident data: same rounds, K and F as the oracle's learner."""
    rng = np.random.default_rng(8)
    eg = family == "ExpGamma"
    N, D = 5000, 6
    if eg:
        rates = rng.uniform(0.1, 8.0, (3, D))
        X = rng.exponential(1.0, (N, D)) / rates[rng.integers(0, 3, N)]
    else:
        mu = rng.normal(0, 6.0, (4, D))
        X = mu[rng.integers(0, 4, N)] + rng.normal(size=(N, D)) * rng.uniform(0.5, 1.5, (1, D))
    with capi.Context(0) as ctx:
        ctx.set_data([X])
        F, model = ctx.cluster(capi.W_DIRICHLET, nthreads=2, ckind=capi.C_EXPGAMMA if eg else capi.C_NORMGAMMA)
        rounds, K = model.rounds(), model.dims()[1]
        q = ctx.get_qz([N])[0]
        model.close()
    tr = []
    Fo, qo, _, clo = (o.learnBEMM if eg else o.learnDGMM)(X, trace=tr)
    assert K == len(clo) and [k for k, _ in rounds] == [k for k, _ in tr]
    for (_, a), (_, b) in zip(rounds, tr):
        np.testing.assert_allclose(a, b, rtol=1e-8)
    assert abs(F - Fo) <= 1e-8 * abs(Fo)
    assert_q_close(q, qo, rtol=1e-6)


@pytest.mark.parametrize("family", ["NormGamma", "ExpGamma"])
@pytest.mark.parametrize("N,D,K,J", [(700, 129, 5, 1), (400, 300, 9, 2), (260, 1000, 3, 1), (150, 257, 70, 3)])
def test_wide_observations_in_the_separable_families(family, N, D, K, J):
    """D > 128 (no limit for NormGamma / ExpGamma): the E-step walks the row tile in chunks of 128 dimensions, the
    statistics run one launch per block of 128 columns; fixed-K VBEM against the oracle, K > 64 included."""
    rng = np.random.default_rng(N + D)
    eg = family == "ExpGamma"
    X, q0 = _data(rng, N, D, K, J, eg)
    cf = o.ExpGamma if eg else o.NormGamma
    wf = o.GDirichlet if J > 1 else o.StickBreak
    wk = capi.W_GDIRICHLET if J > 1 else capi.W_STICKBREAK
    tro, _, qo, _, clo = o.vbem_fixed(X, q0, wf, 1.0, 3, False, cf)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        Nk, xs, xxs, Njk = ctx.suffstat_diag(second=not eg)
        np.testing.assert_allclose(xs, np.vstack(q0).T @ np.vstack(X), rtol=1e-10, atol=1e-9)
        if not eg:
            np.testing.assert_allclose(xxs, np.vstack(q0).T @ (np.vstack(X) ** 2), rtol=1e-10, atol=1e-9)
        F, tr, model = ctx.vbem(wk, fixed_iters=3, ckind=capi.C_EXPGAMMA if eg else capi.C_NORMGAMMA)
        q = ctx.get_qz([x.shape[0] for x in X])
        model.close()
    np.testing.assert_allclose(tr, tro, rtol=1e-9)
    for j in range(J):  # sums over up to 1000 dimensions: absolute agreement to 1e-9 instead of 1e-11
        np.testing.assert_allclose(q[j], qo[j], rtol=1e-7, atol=1e-9)


def test_wide_learners_on_device():
    """Model selection with D = 200 diagonal Gaussians and D = 150 exponentials, end to end against the oracle."""
    import libcluster_amd as lc

    rng = np.random.default_rng(3)
    cent = rng.normal(0, 3.0, (3, 200))
    X = cent[rng.integers(0, 3, 600)] + rng.normal(size=(600, 200))
    tr = []
    Fo, _, _, clo = o.learnDGMM(X, trace=tr)
    F, qZ, w, mu, cov, info = lc.learnDGMM(X, return_info=True)
    assert info["K"] == len(clo) and [k for k, _ in info["rounds"]] == [k for k, _ in tr]
    assert abs(F - Fo) <= 1e-9 * abs(Fo)
    rates = rng.uniform(0.2, 5.0, (2, 150))
    Xe = [rng.exponential(1.0, (300, 150)) / rates[rng.integers(0, 2, 300)] for _ in range(2)]
    tr = []
    Fo, _, _, clo = o.learnEGMC(Xe, trace=tr)
    F, qZ, w, mu, cov, info = lc.learnEGMC(Xe, return_info=True)
    assert info["K"] == len(clo) and [k for k, _ in info["rounds"]] == [k for k, _ in tr]
    assert abs(F - Fo) <= 1e-9 * abs(Fo)
