"""Parity of the HIP path (through the C-ABI) against the oracle and the
committed golden vectors.  Tolerance: the north star asks for qZ and F within
1e-5 relative; the fp64 kernels are held to 1e-9 here."""
import numpy as np
import pytest

import lc_oracle as o
from libcluster_amd import capi

pytestmark = pytest.mark.gpu

RTOL_Q = 1e-9   # relative, on entries with q > 1e-12 (bar: 1e-5)
RTOL_F = 1e-10  # relative on F / Fz (bar: 1e-5)
WF = {"Dirichlet": o.Dirichlet, "StickBreak": o.StickBreak, "GDirichlet": o.GDirichlet}
WK = {"Dirichlet": capi.W_DIRICHLET, "StickBreak": capi.W_STICKBREAK, "GDirichlet": capi.W_GDIRICHLET}


def assert_q_close(got, ref, rtol=RTOL_Q):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    big = ref > 1e-12
    if big.any():
        assert np.max(np.abs(got[big] - ref[big]) / ref[big]) < rtol
    assert np.max(np.abs(got - ref), initial=0.0) < 1e-11


def test_suffstat_matches_golden(estep_cases):
    for c in estep_cases:
        X = [np.array(x) for x in c["X"]]
        q0 = [np.array(q) for q in c["q0"]]
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            smask = None
            if c["sparse"]:
                smask = (np.array(c["Njk"]) >= o.ZEROCUTOFF).astype(np.uint8)
            Nk, xs, xxs, Njk = ctx.suffstat(smask)
        np.testing.assert_allclose(Njk, np.array(c["Njk"]), rtol=1e-12, err_msg=c["name"])
        np.testing.assert_allclose(Nk, c["stats"]["Nk"], rtol=1e-12, err_msg=c["name"])
        np.testing.assert_allclose(xs, np.array(c["stats"]["xs"]), rtol=1e-10, atol=1e-11, err_msg=c["name"])
        np.testing.assert_allclose(xxs, np.array(c["stats"]["xxs"]), rtol=1e-10, atol=1e-9, err_msg=c["name"])
        assert np.array_equal(xxs, np.transpose(xxs, (0, 2, 1)))  # exactly symmetric


def test_estep_matches_golden(estep_cases):
    for c in estep_cases:
        X = [np.array(x) for x in c["X"]]
        p = c["post"]
        active = None
        if c["sparse"]:
            active = (np.array(c["Njk"]) >= o.ZEROCUTOFF).astype(np.uint8)
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            Fz, ll = ctx.estep_posterior(p["nu"], p["beta"], np.array(p["m"]), np.array(p["iW"]), p["logdW"],
                                         np.array(c["Elogpi"]), active)
            q = ctx.get_qz([x.shape[0] for x in X])
        assert abs(Fz - c["Fz1"]) <= RTOL_F * abs(c["Fz1"]), c["name"]
        for j in range(c["J"]):
            assert_q_close(q[j], np.array(c["q1"][j]))
        # data term of the split ordering: sum_n q_nk Eloglike_k(x_n)
        ref = np.zeros(c["K"])
        for j in range(c["J"]):
            ref += np.einsum("nk,nk->k", np.array(c["q1"][j]), np.array(c["Eloglike"][j]))
        np.testing.assert_allclose(ll, ref, rtol=1e-9, atol=1e-9, err_msg=c["name"])


def test_vbem_fixed_matches_golden(estep_cases):
    for c in estep_cases:
        X = [np.array(x) for x in c["X"]]
        q0 = [np.array(q) for q in c["q0"]]
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            F, tr, model = ctx.vbem(WK[c["weights"]], 1.0, c["prior"], sparse=c["sparse"], fixed_iters=c["iters"])
            q = ctx.get_qz([x.shape[0] for x in X])
            np.testing.assert_allclose(tr, c["Ftrace"], rtol=RTOL_F, err_msg=c["name"])
            for j in range(c["J"]):
                assert_q_close(q[j], np.array(c["qT"][j]), rtol=1e-8)
            Ns = [model.cluster(k)["N"] for k in range(c["K"])]
            np.testing.assert_allclose(Ns, c["NkT"], rtol=1e-9, atol=1e-12)
            model.close()


@pytest.mark.parametrize("N,D,K,J", [(1000, 16, 8, 1), (777, 23, 5, 3), (513, 64, 6, 1), (300, 128, 3, 2),
                                      (4099, 2, 2, 1), (50, 7, 33, 1),
                                      # the in-between layouts (48, 80, 96 and 112 columns)
                                      (600, 40, 7, 2), (901, 48, 9, 1), (450, 80, 5, 3), (500, 96, 4, 1),
                                      (333, 70, 6, 1), (512, 110, 3, 2),
                                      # active widths (round 6): 20, 28, 36, 44 columns of the 32- / 48-column layouts, 56, 72, 104 beyond
                                      (700, 18, 6, 1), (650, 27, 11, 2), (800, 35, 5, 1), (500, 42, 14, 1), (600, 53, 7, 1), (480, 66, 4, 2),
                                      (400, 100, 3, 1),
                                      # D = 64 / 80: four row groups per wave where the log q~ table fits in LDS (K = 6 ... 21 / 12), three beyond
                                      (700, 64, 16, 2), (640, 64, 21, 1), (500, 64, 22, 1), (520, 80, 12, 1), (520, 72, 8, 2), (400, 80, 13, 1),
                                      # ragged K: the last cluster slice of the statistics pass runs row-split
                                      (1500, 64, 9, 1), (900, 64, 10, 3), (700, 128, 9, 1), (640, 96, 5, 2),
                                      (800, 32, 17, 1), (4000, 16, 33, 1), (300, 64, 1, 1), (500, 48, 2, 2),
                                      # wider than 128 columns: panel / chunk streaming kernels
                                      (700, 129, 3, 1), (1000, 200, 5, 2), (640, 256, 9, 1), (400, 300, 2, 3)])
def test_estep_and_suffstat_vs_oracle_random(N, D, K, J):
    rng = np.random.default_rng(N + D + K)
    X, q0 = [], []
    for j in range(J):
        n = N // J + (j == 0) * (N % J)
        X.append(rng.normal(size=(n, D)) * 1.5 + rng.integers(0, K, (n, 1)))
        q0.append(rng.dirichlet(np.ones(K) * 0.3, n))
    wf = o.GDirichlet if J > 1 else o.StickBreak
    weights = [wf() for _ in range(J)]
    cl = [o.GaussWish(1.0, D) for _ in range(K)]
    for j in range(J):
        weights[j].update(o.updateSS(X[j], q0[j], cl))
    ref_stats = (np.array([c.N_s for c in cl]), np.stack([c.x_s for c in cl]), np.stack([c.xx_s for c in cl]))
    for c in cl:
        c.update()
    qref, Fzref = [], 0.0
    for j in range(J):
        q, fz = o.vbexpectation(X[j], weights[j], cl)
        qref.append(q)
        Fzref += fz
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        Nk, xs, xxs, Njk = ctx.suffstat()
        np.testing.assert_allclose(Nk, ref_stats[0], rtol=1e-11)
        np.testing.assert_allclose(xs, ref_stats[1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(xxs, ref_stats[2], rtol=1e-9, atol=1e-8)
        Fz, _ = ctx.estep_posterior([c.nu for c in cl], [c.beta for c in cl], np.stack([c.m for c in cl]),
                                    np.stack([c.iW for c in cl]), [c.logdW for c in cl],
                                    np.stack([w.Elogweight() for w in weights]))
        q = ctx.get_qz([x.shape[0] for x in X])
    assert abs(Fz - Fzref) <= RTOL_F * abs(Fzref)
    for j in range(J):
        assert_q_close(q[j], qref[j])
        np.testing.assert_allclose(q[j].sum(axis=1), 1.0, rtol=1e-12)


def test_input_layouts_are_equivalent():
    """Column-major (Eigen default) and row-major X / qZ give identical results."""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 5))
    q = rng.dirichlet(np.ones(3), 200)
    out = []
    for order in ("C", "F"):
        with capi.Context(0) as ctx:
            ctx.set_data(np.array(X, order=order))
            ctx.set_qz(np.array(q, order=order))
            out.append(ctx.suffstat())
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    # several packing blocks per group (the upload packs 1024-row blocks on pool threads), ragged groups, and the
    # way back: rows fetched from the device equal the input whatever its layout was
    sizes = [2500, 0, 1777, 1030]
    Xg = [rng.normal(size=(n, 23)) for n in sizes]
    qg = [rng.dirichlet(np.ones(5), n) for n in sizes]
    res = []
    for order in ("C", "F"):
        with capi.Context(0) as ctx:
            ctx.set_data([np.array(x, order=order) for x in Xg])
            ctx.set_qz([np.array(q, order=order) for q in qg])
            assert np.array_equal(ctx.get_rows(2, 1000, 777), Xg[2][1000:1777])
            for a, b in zip(ctx.get_qz(sizes), qg):
                assert np.array_equal(a, b)
            for a, b in zip(ctx.get_qz_colmajor(sizes), qg):  # the Eigen-layout bulk getter
                assert a.flags.f_contiguous and np.array_equal(a, b)
            res.append(ctx.suffstat())
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_edge_cases():
    rng = np.random.default_rng(9)
    # a single observation, K=1; an empty group among non-empty ones
    with capi.Context(0) as ctx:
        ctx.set_data(np.array([[1.0, 2.0, 3.0]]))
        ctx.fill_qz(1, 1.0)
        Nk, xs, xxs, _ = ctx.suffstat()
        assert Nk[0] == 1.0 and np.array_equal(xs[0], [1.0, 2.0, 3.0])
        np.testing.assert_allclose(xxs[0], np.outer([1, 2, 3], [1, 2, 3]))
    Xg = [rng.normal(size=(20, 3)), np.zeros((0, 3)), rng.normal(size=(33, 3))]
    qg = [rng.dirichlet(np.ones(2), 20), np.zeros((0, 2)), rng.dirichlet(np.ones(2), 33)]
    with capi.Context(0) as ctx:
        ctx.set_data(Xg)
        ctx.set_qz(qg)
        Nk, xs, xxs, Njk = ctx.suffstat()
        np.testing.assert_allclose(Njk[1], 0.0)
        np.testing.assert_allclose(Njk[0], qg[0].sum(axis=0), rtol=1e-13)
        np.testing.assert_allclose(Njk[2], qg[2].sum(axis=0), rtol=1e-13)
    # beyond the widest Gauss-Wishart layout: refused (the separable families accept it, tests/test_gpu_families.py)
    with capi.Context(0) as ctx:
        ctx.set_data(np.zeros((4, 1025)))
        ctx.fill_qz(1, 1.0)
        with pytest.raises(ValueError, match="D > 1024"):
            ctx.suffstat()


def _check_learn(res, ref, rows):
    F, qZ, w, means, covs, info = res
    assert info["K"] == ref["K"]
    assert abs(F - ref["F"]) <= 1e-8 * abs(ref["F"])
    assert [k for k, _ in info["rounds"]] == [k for k, _ in ref["rounds"]]
    for (k, tr), (_, rtr) in zip(info["rounds"], ref["rounds"]):
        np.testing.assert_allclose(tr, rtr, rtol=1e-8)
    np.testing.assert_allclose(info["N"], ref["N"], rtol=1e-7)
    np.testing.assert_allclose(np.vstack(means), np.array(ref["means"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(covs), np.array(ref["covs"]), rtol=1e-7, atol=1e-9)
    assert means[0].shape == (1, len(ref["means"][0])) and np.asarray(w[0] if isinstance(w, list) else w).shape[1] == 1
    qs = qZ if isinstance(qZ, list) else [qZ]
    for a, b in zip(qs, ref["qZ"]):
        assert_q_close(a, np.array(b), rtol=1e-6)
    el = info["Elogweight"]
    np.testing.assert_allclose(np.array(el), np.array(ref["Elogweight"]), rtol=1e-8)


def test_learnBGMM_on_reference_test_data(xcat, xcat_traces):
    """BASELINE config 1: learnBGMM on test/testdata.h Xcat."""
    import libcluster_amd as lc

    _check_learn(lc.learnBGMM(xcat["Xcat"], return_info=True), xcat_traces["learnBGMM"], [120])
    F, *_, info = lc.learnBGMM(xcat["Xcat"], maxclusters=1, return_info=True)
    assert info["K"] == 1 and abs(F - xcat_traces["learnBGMM_max1"]["F"]) < 1e-8


def test_learnVDP_on_reference_test_data(xcat, xcat_traces):
    import libcluster_amd as lc

    _check_learn(lc.learnVDP(xcat["Xcat"], return_info=True), xcat_traces["learnVDP"], [120])
    F, *_, info = lc.learnVDP(xcat["Xcat"], concentration=2.5, return_info=True)
    assert info["K"] == xcat_traces["learnVDP_conc2.5"]["K"]
    assert abs(F - xcat_traces["learnVDP_conc2.5"]["F"]) < 1e-7


def test_learnGMC_on_reference_test_data(xcat, xcat_traces):
    """The reference's own test main: test/cluster_test.cpp:38-69."""
    import libcluster_amd as lc

    _check_learn(lc.learnGMC(xcat["X"], return_info=True), xcat_traces["learnGMC"], [10] * 12)
    ref = xcat_traces["learnGMC_sparse"]
    if ref["throws"]:
        with pytest.raises(RuntimeError, match="Free energy increase"):
            lc.learnGMC(xcat["X"], sparse=True)


def test_learn_argument_errors(xcat):
    import libcluster_amd as lc

    with pytest.raises(ValueError):
        lc.learnBGMM(xcat["Xcat"], nthreads=0)  # cluster.cpp:576-577
    with pytest.raises(ValueError):
        lc.learnBGMM(xcat["Xcat"], prior=-1.0)  # distributions.cpp:282-283
    with pytest.raises(ValueError):
        lc.learnVDP(xcat["Xcat"], concentration=0.0)  # distributions.cpp:107-108


def test_eloglike_matches_oracle():
    """GaussWish::Eloglike (distributions.cpp:356-370) through lc_eloglike."""
    rng = np.random.default_rng(11)
    N, D, K = 333, 7, 3
    X = rng.normal(size=(N, D)) * 2 + 1
    q = rng.dirichlet(np.ones(K), N)
    cl = [o.GaussWish(0.7, D) for _ in range(K)]
    o.updateSS(X, q, cl)
    for c in cl:
        c.update()
    ref = np.stack([c.Eloglike(X) for c in cl], axis=1)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        got = ctx.eloglike([c.nu for c in cl], [c.beta for c in cl], np.stack([c.m for c in cl]),
                           np.stack([c.iW for c in cl]), [c.logdW for c in cl], [N])[0]
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_learnSGMC_on_reference_test_data(xcat, xcat_traces):
    """learnSGMC (include/libcluster.h:409-419): Dirichlet weights per group, same kernels."""
    import libcluster_amd as lc

    _check_learn(lc.learnSGMC(xcat["X"], return_info=True), xcat_traces["learnSGMC"], [10] * 12)


def test_cluster_on_device_resident_data_matches_oracle():
    """Model selection end to end on data that never leaves the GPU (lc_ctx_synth + lc_cluster):
    the split search's partobs / splitobs / auglabels run on the device.  Same rounds, K and F as the oracle."""
    rng = np.random.default_rng(21)
    N, D, Kt = 6000, 5, 4
    mu = rng.normal(0, 6.0, (Kt, D))
    L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D))))
                  for _ in range(Kt)])
    with capi.Context(0) as ctx:
        ctx.synth(N, D, Kt, mu, L, 77, 0, 0.9)
        X = ctx.get_rows(0, 0, N)
        F, model = ctx.cluster(capi.W_STICKBREAK, nthreads=4)
        rounds = model.rounds()
        K = model.dims()[1]
        q = ctx.get_qz([N])[0]
        model.close()
    tr = []
    Fo, qo, _, clo = o.learnVDP(X, trace=tr)
    assert K == len(clo) and [k for k, _ in rounds] == [k for k, _ in tr]
    for (_, a), (_, b) in zip(rounds, tr):
        np.testing.assert_allclose(a, b, rtol=1e-8)
    assert abs(F - Fo) <= 1e-8 * abs(Fo)
    assert_q_close(q, qo, rtol=1e-6)


def test_random_shapes_sweep():
    """Many small random shapes (D = 1..128, K = 1..70, J = 1..5, ragged and tiny groups, rows not a
    multiple of 16/32/256) through suff-stats + E-step + one fixed-K VBEM iteration, against the oracle."""
    rng = np.random.default_rng(20261001)
    for trial in range(40):
        D = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 47, 64, 65, 100, 128]))
        K = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 31, 33, 70]))
        J = int(rng.integers(1, 6))
        Ns = [int(rng.choice([1, 2, 15, 16, 17, 31, 33, 100, 257, 600])) for _ in range(J)]
        if sum(Ns) * K * D > 6_000_000:
            Ns = [min(n, 64) for n in Ns]
        X = [rng.normal(size=(n, D)) * 1.3 + rng.integers(0, 3, (n, 1)) for n in Ns]
        q0 = [rng.dirichlet(np.ones(K) * 0.5, n) for n in Ns]
        wname = ["Dirichlet", "StickBreak", "GDirichlet"][trial % 3]
        Ftr, Fztr, qT, _, clT = o.vbem_fixed(X, q0, WF[wname], 1.0, 1)
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            F, tr, model = ctx.vbem(WK[wname], fixed_iters=1)
            q = ctx.get_qz(Ns)
            Ng = [model.cluster(k)["N"] for k in range(K)]
            model.close()
        msg = f"trial {trial}: D={D} K={K} Ns={Ns} {wname}"
        assert abs(tr[0] - Ftr[0]) <= 1e-9 * abs(Ftr[0]), msg
        np.testing.assert_allclose(Ng, [c.getN() for c in clT], rtol=1e-10, atol=1e-12, err_msg=msg)
        for j in range(J):
            assert_q_close(q[j], qT[j], rtol=1e-8)


def test_many_clusters_and_many_small_groups():
    """K well above the register-resident paths (K = 150: cluster slices, re-read normalisation) and J = 1500 tiny
    ragged groups (block-per-group column sums, per-group constant tables), all three cluster families."""
    rng = np.random.default_rng(77)
    # (a) many clusters
    N, D, K = 3000, 6, 150
    X = [rng.normal(size=(N, D)) * 2 + rng.integers(0, 12, (N, 1))]
    q0 = [rng.dirichlet(np.ones(K) * 0.2, N)]
    for cf, ck in ((o.GaussWish, capi.C_GAUSSWISH), (o.NormGamma, capi.C_NORMGAMMA), (o.ExpGamma, capi.C_EXPGAMMA)):
        Xs = [np.abs(X[0]) + 0.05] if cf is o.ExpGamma else X
        tro, _, qo, _, clo = o.vbem_fixed(Xs, q0, o.StickBreak, 1.0, 2, False, cf)
        with capi.Context(0) as ctx:
            ctx.set_data(Xs)
            ctx.set_qz(q0)
            F, tr, model = ctx.vbem(capi.W_STICKBREAK, fixed_iters=2, ckind=ck, nthreads=4)
            q = ctx.get_qz([N])
            model.close()
        np.testing.assert_allclose(tr, tro, rtol=1e-9, err_msg=cf.__name__)
        assert_q_close(q[0], qo[0], rtol=1e-7)
    # (b) many small groups
    J, K = 1500, 5
    sizes = rng.integers(1, 40, J)
    sizes[7] = 0
    Xg = [rng.normal(size=(int(n), 3)) + rng.integers(0, K, (int(n), 1)) * 3.0 for n in sizes]
    qg = [rng.dirichlet(np.ones(K) * 0.5, int(n)) for n in sizes]
    for cf, ck in ((o.GaussWish, capi.C_GAUSSWISH), (o.NormGamma, capi.C_NORMGAMMA)):
        tro, _, qo, wo, _ = o.vbem_fixed(Xg, qg, o.GDirichlet, 1.0, 2, False, cf)
        with capi.Context(0) as ctx:
            ctx.set_data(Xg)
            ctx.set_qz(qg)
            F, tr, model = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=2, ckind=ck)
            q = ctx.get_qz([int(n) for n in sizes])
            el = np.stack([model.weights(j)[0] for j in (0, 7, 733, J - 1)])
            model.close()
        np.testing.assert_allclose(tr, tro, rtol=1e-9, err_msg=cf.__name__)
        np.testing.assert_allclose(el, np.stack([wo[j].Elogweight() for j in (0, 7, 733, J - 1)]), rtol=1e-9)
        for j in range(0, J, 97):
            assert_q_close(q[j], qo[j], rtol=1e-7)


def test_sparse_mode_and_zero_skipping_are_exact():
    """Grouped data where every group uses a few of the clusters: the sparse variants (waves skip clusters that are
    inactive for all their rows; statistics skip all-zero steps) give the oracle's sparse results, and the
    zero-skipping statistics pass equals the dense one bit for bit."""
    rng = np.random.default_rng(5)
    J, K, D = 6, 8, 5
    X, q0 = [], []
    for j in range(J):
        n = 900 + 64 * j
        use = rng.choice(K, 2, replace=False)
        z = rng.choice(use, n)
        X.append(rng.normal(size=(n, D)) + 9.0 * np.eye(K, D)[z] * 1.0 + z[:, None])
        q = np.zeros((n, K))
        q[np.arange(n), z] = 1.0  # hard start: most responsibilities are exactly zero
        q0.append(q)
    tro, _, qo, wo, clo = o.vbem_fixed(X, q0, o.GDirichlet, 1.0, 3, True)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q0)
        dense = ctx.suffstat()
        ctx.set_skip_zero(True)
        skipped = ctx.suffstat()
        for a, b in zip(dense, skipped):
            np.testing.assert_array_equal(a, b)
        F, tr, model = ctx.vbem(capi.W_GDIRICHLET, sparse=True, fixed_iters=3)
        q = ctx.get_qz([x.shape[0] for x in X])
        model.close()
    np.testing.assert_allclose(tr, tro, rtol=1e-10)
    for j in range(J):
        assert_q_close(q[j], qo[j], rtol=1e-8)

    # the work-list path with several cluster slices per group, an empty group, a one-row group and a group that uses
    # most clusters: statistics with a mask equal the masked dense sums
    J, K, D = 7, 40, 9
    sizes = [700, 0, 1, 333, 2048, 90, 1500]
    X = [rng.normal(size=(n, D)) * 2 for n in sizes]
    qz = [rng.dirichlet(np.ones(K) * 0.3, n) if n else np.zeros((0, K)) for n in sizes]
    mask = (rng.random((J, K)) < 0.2).astype(np.uint8)
    mask[4] = 1
    mask[4, ::7] = 0
    mask[5] = 0
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(qz)
        Nk, xs, xxs, Njk = ctx.suffstat(mask)
    Nr, xr, xxr = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D))
    for j in range(J):
        for k in range(K):
            if mask[j, k] and sizes[j]:
                w = qz[j][:, k]
                Nr[k] += w.sum()
                xr[k] += w @ X[j]
                xxr[k] += (X[j] * w[:, None]).T @ X[j]
    np.testing.assert_allclose(Nk, Nr, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(xs, xr, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(xxs, xxr, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(Njk, np.stack([q.sum(axis=0) for q in qz]), rtol=1e-11, atol=1e-13)


def test_randomised_parity_fuzz():
    """tools/fuzz_parity.py: random shapes, group structures, weight kinds, cluster families, sparse on/off, hard and
    soft starts (hard starts produce tied counts, which exercise std::sort's tie order for K > 16)."""
    import subprocess
    import sys
    from pathlib import Path

    import os

    root = Path(__file__).resolve().parents[1]
    n = os.environ.get("LC_FUZZ_CASES", "30")  # (the suite's share; tools/fuzz_*.py run thousands per round: profiles/rNN_fuzz.log)
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_parity.py"), n, "7"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{n} cases" in r.stdout and " 0 failures" in r.stdout


def test_randomised_learner_fuzz():
    """tools/fuzz_learn.py: full model selection (all ten learners, random data / priors / maxclusters / sparse /
    document structures) -- every round's K and free energies and the final K, T and F against the oracle."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_learn.py"), "150", "11"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "150 cases" in r.stdout and " 0 failures" in r.stdout


def test_randomised_kernel_fuzz():
    """tools/fuzz_kernels.py: statistics and E-step of the three families at mid sizes (up to 400k rows, D to 128, K to
    100, 1-31 ragged groups) against numpy / the oracle on row subsets: chunk, slice, batch and row-group edges."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_kernels.py"), "60", "21"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 cases" in r.stdout and " 0 failures" in r.stdout


@pytest.mark.parametrize("D", [2, 5, 8, 9])
def test_fused_pass_on_hard_labels_after_a_split(D):
    """One VBEM iteration from hard labels with one cluster cut in two along its principal axis (what model selection
    hands the fused pass): at D <= 8 this is the half-width instance, whose last lane-sum link sits right in front of
    the row maximum -- an MFMA result read too early there once gave NaN rows in this very case."""
    rng = np.random.default_rng(3)
    K, N = 3, 2000
    mu = rng.normal(0, 5, (K, D))
    z = rng.integers(0, K, N)
    X = mu[z] + rng.normal(size=(N, D))
    idx = np.flatnonzero(z == 0)
    Xk = X[idx] - X[idx].mean(0)
    v = np.linalg.eigh(np.atleast_2d(np.cov(Xk.T)))[1][:, -1]
    qa = np.zeros((N, K + 1))
    qa[np.arange(N), z] = 1.0
    mv = idx[Xk @ v < 0]
    qa[mv, K], qa[mv, 0] = 1.0, 0.0
    w = o.StickBreak()
    cl = [o.GaussWish(1.0, D) for _ in range(K + 1)]
    w.update(o.updateSS(X, qa, cl))
    for c in cl:
        c.update()
    qref, _ = o.vbexpectation(X, w, cl)
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(qa)
        F, tr, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=1)
        q1 = ctx.get_qz([N])[0]
        Nk, xs, xxs, _ = ctx.suffstat()
        m.close()
    assert np.isfinite(q1).all() and np.isfinite(tr).all()
    assert_q_close(q1, qref)
    np.testing.assert_allclose(Nk, qref.sum(0), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(xs, qref.T @ X, rtol=1e-9, atol=1e-9)


def test_half_width_fused_instance_equals_the_full_width_one():
    """D <= 8 takes fused_small_kernel's half-width instance; LC_FUSED_FULL=1 (read once per process) takes the full-width
    one on the same padded layout.  The columns left out are zeros, so both must give the same F trace, responsibilities
    and statistics to rounding (in practice: to the last bit)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    child = (
        "import sys, json, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "from libcluster_amd import capi\n"
        "out = []\n"
        "for D, K, N, J in ((3, 4, 5000, 1), (8, 16, 7001, 1), (5, 7, 4000, 3)):\n"
        "    rng = np.random.default_rng(D * 100 + K)\n"
        "    X = [rng.normal(size=(N // J, D)) * 1.3 + rng.integers(0, K, (N // J, 1)) for _ in range(J)]\n"
        "    q0 = [rng.dirichlet(np.ones(K) * 0.4, x.shape[0]) for x in X]\n"
        "    with capi.Context(0) as ctx:\n"
        "        ctx.set_data(X); ctx.set_qz(q0)\n"
        "        F, tr, m = ctx.vbem(capi.W_GDIRICHLET if J > 1 else capi.W_STICKBREAK, fixed_iters=6)\n"
        "        q = ctx.get_qz([x.shape[0] for x in X])\n"
        "        Nk, xs, xxs, _ = ctx.suffstat()\n"
        "        m.close()\n"
        "    out.append([tr.tolist(), [float(np.sum(a * np.arange(1, a.size + 1).reshape(a.shape))) for a in q], Nk.tolist(), float(xs.sum()), float(xxs.sum())])\n"
        "print('RESULT', json.dumps(out))\n"
    )
    res = []
    for full in (False, True):
        env = dict(os.environ)
        env.pop("LC_FUSED_FULL", None)
        if full:  # (a switch of the test-hooks build: lck::test_switch)
            env["LC_FUSED_FULL"] = "1"
            env["LC_LIB_PATH"] = str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[7:]))
    for half, fullw in zip(*res):
        np.testing.assert_allclose(half[0], fullw[0], rtol=1e-13)
        np.testing.assert_allclose(half[1], fullw[1], rtol=1e-12)
        np.testing.assert_allclose(half[2], fullw[2], rtol=1e-12)
        np.testing.assert_allclose(half[3:], fullw[3:], rtol=1e-11)


def test_register_resident_wide_estep_equals_the_streaming_one_bit_for_bit():
    """D = 129 ... 256 takes estep_wide_kernel's instances that keep the rows' X fragments in registers (three / four column
    panels); LC_WIDE_STREAM=1 (test-hooks build) takes the streaming instance that serves every width.  Same tiles, same
    order of MFMAs per accumulator: F traces and responsibilities must agree to the last bit."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    child = (
        "import sys, json, hashlib, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "from libcluster_amd import capi\n"
        "out = []\n"
        "for D, K, N, J in ((129, 3, 700, 1), (192, 5, 1301, 1), (200, 4, 900, 3), (256, 6, 1500, 1)):\n"
        "    rng = np.random.default_rng(D * 100 + K)\n"
        "    X = [rng.normal(size=(N // J, D)) * 1.3 + rng.integers(0, K, (N // J, 1)) for _ in range(J)]\n"
        "    q0 = [rng.dirichlet(np.ones(K) * 0.4, x.shape[0]) for x in X]\n"
        "    with capi.Context(0) as ctx:\n"
        "        ctx.set_data(X); ctx.set_qz(q0)\n"
        "        F, tr, m = ctx.vbem(capi.W_GDIRICHLET if J > 1 else capi.W_STICKBREAK, fixed_iters=4)\n"
        "        q = ctx.get_qz([x.shape[0] for x in X])\n"
        "        m.close()\n"
        "    out.append([[float.hex(float(f)) for f in tr], [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in q]])\n"
        "print('RESULT', json.dumps(out))\n"
    )
    res = []
    for stream in (False, True):
        env = dict(os.environ)
        env.pop("LC_WIDE_STREAM", None)
        env["LC_LIB_PATH"] = str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
        if stream:  # (a switch of the test-hooks build: lck::test_switch)
            env["LC_WIDE_STREAM"] = "1"
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[7:]))
    assert res[0] == res[1]


def test_mahaldist_matches_numpy():
    """probutils::mahaldist (probutils.cpp:113-138) on the GPU: ragged groups, narrow and wide D, SPD A; non-PD is
    refused."""
    rng = np.random.default_rng(12)
    for D, sizes in ((3, [50, 0, 7]), (23, [400]), (64, [1000, 33]), (128, [257]), (190, [300, 5]), (320, [200])):
        X = [rng.normal(size=(n, D)) * 2 + 1 for n in sizes]
        B = rng.normal(size=(D, D))
        A = B @ B.T / D + 0.3 * np.eye(D)
        mu = rng.normal(size=D)
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            d2 = ctx.mahaldist(mu, A)
            with pytest.raises(ValueError, match="not positive definite"):
                ctx.mahaldist(mu, A - 5.0 * np.eye(D))
        Xa = np.vstack(X) - mu
        ref = np.einsum("nd,nd->n", Xa, np.linalg.solve(A, Xa.T).T)
        np.testing.assert_allclose(d2, ref, rtol=1e-10, atol=1e-10)


def test_wide_gauss_wishart_learners():
    """Full-covariance model selection on observations wider than 128 columns (estep_wide_kernel, the panel
    launches of suffstat_kernel): same rounds, K, F and responsibilities as the oracle."""
    import libcluster_amd as lc

    rng = np.random.default_rng(33)
    D = 140
    X = np.vstack([rng.normal(size=(260, D)) + 4.0, rng.normal(size=(240, D)) * 0.7 - 3.0])
    X = X[rng.permutation(len(X))]
    tr = []
    Fo, qo, wo, clo = o.learnBGMM(X, trace=tr)
    F, qZ, w, mu, cov, info = lc.learnBGMM(X, return_info=True)
    assert info["K"] == len(clo) and [k for k, _ in info["rounds"]] == [k for k, _ in tr]
    assert abs(F - Fo) <= 1e-8 * abs(Fo)
    assert_q_close(qZ, qo, rtol=1e-6)
    np.testing.assert_allclose(np.vstack(mu), np.stack([c.getmean() for c in clo]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(cov), np.stack([c.getcov() for c in clo]), rtol=1e-7, atol=1e-9)
    # grouped, sparse
    Xg = [X[:200], X[200:330], X[330:]]
    tr = []
    Fo, qo, wo, clo = o.learnGMC(Xg, trace=tr)
    F, qZ, w, mu, cov, info = lc.learnGMC(Xg, return_info=True)
    assert info["K"] == len(clo) and abs(F - Fo) <= 1e-8 * abs(Fo)
    for a, b in zip(qZ, qo):
        assert_q_close(a, b, rtol=1e-6)


@pytest.mark.gpu
def test_context_takes_new_observations_with_more_rows(lib):
    """A context that has worked on N rows is handed 40x as many (lc_ctx_set_data again): every row-sized buffer must
    follow (the responsibility buffers once kept their old size)."""
    from libcluster_amd import capi

    rng = np.random.default_rng(12)
    res = []
    for reuse in (False, True):
        with capi.Context(0) as ctx:
            if reuse:
                Xs = rng.normal(size=(500, 24))
                ctx.set_data(Xs)
                ctx.set_qz(np.random.default_rng(1).dirichlet(np.ones(7), 500))
                ctx.vbem(capi.W_DIRICHLET, fixed_iters=2)[2].close()
                F0, m0 = ctx.cluster(capi.W_STICKBREAK)
                m0.close()
            r2 = np.random.default_rng(13)
            X = r2.normal(size=(20000, 24)) + 3.0 * r2.integers(0, 3, (20000, 1))
            q = r2.dirichlet(np.ones(7), 20000)
            ctx.set_data(X)
            ctx.set_qz(q)
            F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=3)
            m.close()
            Fc, mc = ctx.cluster(capi.W_STICKBREAK)
            K = mc.dims()[1]
            mc.close()
            res.append((tr, ctx.get_qz([20000])[0], Fc, K))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]


@pytest.mark.gpu
@pytest.mark.parametrize("wkind,wname", [(capi.W_DIRICHLET, "Dirichlet"), (capi.W_STICKBREAK, "StickBreak")])
def test_prune_clusters_drops_columns_and_updates_weights_like_the_reference(lib, wkind, wname):
    """prune_clusters (cluster.cpp:505-552) through lc_prune: clusters that had no observations in the last M-step go
    (model and qZ columns, the survivors' columns untouched and NOT renormalised), the weights are updated from the
    remaining columns' sums."""
    rng = np.random.default_rng(21)
    N, D, K = 700, 5, 6
    X = rng.normal(size=(N, D)) + 4.0 * rng.integers(0, 3, (N, 1))
    q = np.zeros((N, K))
    q[:, [0, 2, 5]] = rng.dirichlet(np.ones(3), N)  # clusters 1, 3 and 4 start (and stay, in the M-step) empty
    _, _, qo, wo, co = o.vbem_fixed([X], [q], getattr(o, wname), 1.0, 1)
    assert o.prune_clusters(qo, wo, co) and len(co) == 3
    with capi.Context(0) as ctx:
        ctx.set_data(X)
        ctx.set_qz(q)
        F, tr, m = ctx.vbem(wkind, fixed_iters=1)
        assert ctx.prune(m) == 3
        assert m.dims()[1] == 3
        got = ctx.get_qz([N])[0]
        np.testing.assert_allclose(got, qo[0], rtol=1e-10, atol=1e-300)
        elog, nk = m.weights(0)
        np.testing.assert_allclose(nk, wo[0].getNk(), rtol=1e-12)
        np.testing.assert_allclose(elog, wo[0].Elogweight(), rtol=1e-10)
        for k in range(3):
            np.testing.assert_allclose(m.cluster(k)["N"], co[k].getN(), rtol=1e-12)
        assert ctx.prune(m) == 0  # nothing left to drop
        # the pruned model carries on: another iteration equals the oracle's on its pruned state
        F2, tr2, m = ctx.vbem(wkind, fixed_iters=1, model=m)
        Fo, _, _, _, _ = o.vbem_fixed([X], [qo[0]], getattr(o, wname), 1.0, 1)
        np.testing.assert_allclose(tr2[0], Fo[0], rtol=1e-10)
        m.close()


@pytest.mark.gpu
def test_mstep_failure_on_a_pool_thread_reaches_the_caller(lib):
    """A NaN observation poisons one cluster's scatter matrix: GaussWish::update's log-determinant throws
    (distributions.cpp:335-336) on whichever pool thread runs that cluster, and the caller sees the reference's
    runtime_error -- also on the next call (the pool survives)."""
    rng = np.random.default_rng(31)
    N, D, K = 4000, 64, 8
    X = rng.normal(size=(N, D)) + 2.0 * rng.integers(0, K, (N, 1))
    q = rng.dirichlet(np.ones(K), N)
    with capi.Context(0) as ctx:
        Xbad = X.copy()
        Xbad[17, 3] = np.nan
        ctx.set_data(Xbad)
        ctx.set_qz(q)
        for _ in range(2):
            with pytest.raises(RuntimeError, match="not positive definite"):
                ctx.vbem(capi.W_DIRICHLET, fixed_iters=2, nthreads=8)
            ctx.set_qz(q)
        ctx.set_data(X)
        ctx.set_qz(q)
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=2, nthreads=8)
        m.close()
        assert np.isfinite(F)


def test_feature_gemm_statistics_at_every_width_and_cluster_range(lib):
    """suffstat_feat_kernel (the statistics as ONE feature GEMM, cluster index inside the MFMA) forced on wherever it
    exists (LC_SS_FEAT=2: D = 17 ... 128, more than 16 clusters, cluster ranges of <= 32 per launch) and the per-cluster
    suffstat_kernel (LC_SS_FEAT=0) on the same ragged inputs: N_k, s_k, S_k of updateSS / GaussWish::addobs
    (src/cluster.cpp:53-82, src/distributions.cpp:301-313) against numpy to 1e-12 of the largest entry, S_k exactly
    symmetric."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for mode in ("2", "0"):
        e = dict(os.environ, LC_SS_FEAT=mode, LC_SSFEAT_NOTIME="1",
                 LC_LIB_PATH=str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so"))  # (lck::test_switch)
        r = subprocess.run([sys.executable, str(root / "tools" / "ssfeat_check.py"), "child"], capture_output=True, text=True,
                           env=e, timeout=900, cwd=str(root))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        cases, _ = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert len(cases) >= 20
        for c in cases:
            assert c["sym"], (mode, c)
            assert c["eN"] < 1e-12 and c["es"] < 1e-12 and c["eS"] < 1e-12, (mode, c)



@pytest.mark.gpu
def test_quad_statistics_kernel_at_every_instance(lib):
    """suffstat_quad_kernel (K <= 16, D <= 64: the four clusters of a quad in the four blocks of one MFMA, round 6) on ragged
    inputs at every instance -- 1, 2 and 4 parts per quad, both active widths of the 32-, 48- and 64-column layouts -- and,
    with LC_SS_QUAD=0 (test-hooks library), the kernels it replaces on the same inputs: N_k, s_k, S_k of updateSS /
    GaussWish::addobs (src/cluster.cpp:53-82, src/distributions.cpp:301-313) against numpy to 1e-12 of the largest entry,
    S_k exactly symmetric."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sums = []
    for mode in ("1", "0"):
        e = dict(os.environ, LC_SS_QUAD=mode, LC_SSFEAT_NOTIME="1", LC_SSCHECK_CASES="quad",
                 LC_LIB_PATH=str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so"))  # (lck::test_switch)
        r = subprocess.run([sys.executable, str(root / "tools" / "ssfeat_check.py"), "child"], capture_output=True, text=True,
                           env=e, timeout=900, cwd=str(root))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        cases, _ = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert len(cases) >= 20
        for c in cases:
            assert c["sym"], (mode, c)
            assert c["eN"] < 1e-12 and c["es"] < 1e-12 and c["eS"] < 1e-12, (mode, c)
        sums.append([c["h"] for c in cases])
    np.testing.assert_allclose(sums[0], sums[1], rtol=1e-11)


_R4_SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
from libcluster_amd import capi
rng = np.random.default_rng({seed})
N, D, K = {N}, {D}, {K}
mu = rng.normal(0, 1.2, (K, D))
z = rng.integers(0, K, N)
X = mu[z] + rng.normal(size=(N, D))
q0 = rng.dirichlet(np.ones(K) * 0.5, N)
with capi.Context(0) as ctx:
    ctx.set_data(X)
    ctx.set_qz(q0)
    F, tr, model = ctx.vbem(capi.W_STICKBREAK, 1.0, 1.0, fixed_iters=3)
    q = ctx.get_qz([N])[0]
    model.close()
np.save({out!r}, q)
print(json.dumps(dict(F=[float(v) for v in tr])))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,K", [(150_000, 64, 16), (120_000, 57, 8), (100_000, 80, 12), (90_000, 72, 7), (100_001, 64, 21)])
def test_four_row_group_estep_at_64_and_80_columns_agrees_with_the_three_row_group_one(N, D, K, tmp_path):
    """D = 64 / 80 with 6 ... 21 / 12 clusters run estep_kernel's four-row-group scheme (log q~ table in LDS, one
    exponential per entry, selector-chain epilogue: lc_kernels_estep.hip, estep_four_groups) since round 6; LC_ES_R4=0
    (test-hooks library) keeps three row groups per wave.  On inputs far beyond the oracle's reach the two must agree to
    rounding (one against two exponentials per entry: a few ulp in q) over three whole iterations -- both are held to the
    oracle at small sizes by the random-shape test above."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    hooked = str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
    res = []
    for tag, env in (("r4", {}), ("r3", {"LC_LIB_PATH": hooked, "LC_ES_R4": "0"})):
        out = str(tmp_path / f"q_{tag}.npy")
        r = subprocess.run([sys.executable, "-c", _R4_SNIPPET.format(root=str(root), seed=N + D + K, N=N, D=D, K=K, out=out)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=str(root))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append((json.loads(r.stdout.strip().splitlines()[-1])["F"], np.load(out)))
    (Fa, qa), (Fb, qb) = res
    np.testing.assert_allclose(Fa, Fb, rtol=1e-13)
    big = qb > 1e-200
    assert np.max(np.abs(qa[big] - qb[big]) / qb[big]) < 1e-11
    assert not np.array_equal(qa, qb) or K < 6  # (two different sweeps: if every bit agreed the switch would not have switched)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,K", [(120_000, 23, 16), (150_000, 18, 8), (100_000, 40, 20), (90_000, 56, 8), (60_000, 72, 24), (80_000, 35, 5),
                                   (50_000, 104, 12)])
def test_active_width_agrees_with_the_padded_width(N, D, K, tmp_path):
    """The E-step and the statistics kernels walk the 4-column tiles of D rounded up to 8 (to 4 up to 48 columns), not those
    of the 16-column padding (lc_kernels.h, estep_active_width; round 6).  LC_FULL_WIDTH (test-hooks library) makes every
    kernel walk the padded width as round 5 did: the skipped products are 0 x 0, so three whole iterations must agree to
    the rounding of sums taken over another number of row chunks -- on inputs far beyond the oracle's reach (the oracle
    holds both at small sizes)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    hooked = str(root / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
    res = []
    for tag, env in (("active", {}), ("padded", {"LC_LIB_PATH": hooked, "LC_FULL_WIDTH": "1"})):
        out = str(tmp_path / f"q_{tag}.npy")
        r = subprocess.run([sys.executable, "-c", _R4_SNIPPET.format(root=str(root), seed=N + D + K, N=N, D=D, K=K, out=out)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=str(root))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append((json.loads(r.stdout.strip().splitlines()[-1])["F"], np.load(out)))
    (Fa, qa), (Fb, qb) = res
    np.testing.assert_allclose(Fa, Fb, rtol=1e-12)
    big = qb > 1e-200
    assert np.max(np.abs(qa[big] - qb[big]) / qb[big]) < 1e-9
