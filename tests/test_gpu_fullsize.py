"""BASELINE.json's full sizes on the GPU, checked through size-independent
properties (the oracle cannot run 10M x 64 x 32 in seconds):
  * rows of qZ sum to 1  <=>  sum_k N_k = N, and per sampled row;
  * the E-step is row-independent given the posterior: the first rows of the
    full run equal, bit for bit, a separate run on just those rows, which in
    turn equals the oracle;
  * suff-stats are exactly symmetric;
  * statistics and sum_n logZ are additive over row shards (two half-size contexts generated from
    the same Philox stream reproduce the unsharded values -- the multi-GPU invariant);
  * the free energy never increases over EM iterations.
"""
import numpy as np
import pytest

import lc_oracle as o
from libcluster_amd import capi

pytestmark = pytest.mark.gpu


def _mixture(D, K, seed):
    rng = np.random.default_rng(seed)
    mu = rng.normal(0.0, 3.0, (K, D))
    L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D))))
                  for _ in range(K)])
    return mu, L


@pytest.mark.parametrize("N,D,K,seed", [(1_000_000, 16, 8, 1002), (10_000_000, 64, 32, 1003)])
def test_full_size_properties(N, D, K, seed):
    mu, L = _mixture(D, K, seed)
    P = 4096  # prefix checked against the oracle
    with capi.Context(0) as ctx:
        ctx.synth(N, D, K, mu, L, seed, 0, 0.9)
        Xp = ctx.get_rows(0, 0, P)
        q0p = ctx.get_qz_rows(0, 0, P)
        np.testing.assert_allclose(q0p.sum(axis=1), 1.0, rtol=1e-13)
        Nk, xs, xxs, Njk = ctx.suffstat()
        assert abs(Nk.sum() - N) <= 1e-9 * N                       # rows of the initial qZ sum to 1
        assert np.array_equal(xxs, np.transpose(xxs, (0, 2, 1)))   # exactly symmetric
        np.testing.assert_array_equal(Njk[0], Nk)
        assert np.all(np.einsum("kii->k", xxs) > 0)
        # posterior from the full statistics (host M-step through the C-ABI)
        post = [capi.gw_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
        elog, _ = capi.weights_update(capi.W_STICKBREAK, Nk)
        args = ([p["nu"] for p in post], [p["beta"] for p in post], np.stack([p["m"] for p in post]),
                np.stack([p["iW"] for p in post]), [p["logdW"] for p in post], elog[None, :])
        Fz, _ = ctx.estep_posterior(*args, want_ll=False)
        qp = ctx.get_qz_rows(0, 0, P)
        qmid = ctx.get_qz_rows(0, N // 2 - 7, 1001)
        colsum = ctx.colsums()[0]
        F, tr, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=3, nthreads=16)
        m.close()
    assert abs(colsum.sum() - N) <= 1e-9 * N                       # every row of the new qZ sums to 1
    np.testing.assert_allclose(qp.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(qmid.sum(axis=1), 1.0, rtol=1e-12)
    assert np.all(np.diff(tr) <= 1e-9 * abs(tr[0]))                # F is non-increasing

    # the same posterior on the prefix alone: identical rows, and equal to the oracle
    with capi.Context(0) as c2:
        c2.set_data(Xp)
        Fzp, _ = c2.estep_posterior(*args, want_ll=False)
        qp2 = c2.get_qz([P])[0]
    np.testing.assert_array_equal(qp, qp2)
    cl = []
    for k in range(K):
        g = o.GaussWish(1.0, D)
        g.nu, g.beta, g.m, g.iW, g.logdW = (post[k]["nu"], post[k]["beta"], post[k]["m"], post[k]["iW"],
                                            post[k]["logdW"])
        cl.append(g)
    w = o.StickBreak()
    w.E_logpi, w.Nk = elog, Nk
    qref, Fzref = o.vbexpectation(Xp, w, cl)
    big = qref > 1e-12
    assert np.max(np.abs(qp[big] - qref[big]) / qref[big]) < 1e-9
    assert abs(Fzp - Fzref) <= 1e-10 * abs(Fzref)
    assert np.isfinite(Fz) and Fz > 0

    # row sharding (what the multi-GPU path does): statistics and Fz of two half-shards generated
    # from the same Philox stream add up to the unsharded values
    half = N // 2
    tot_stats, tot_Fz = None, 0.0
    for r in range(2):
        with capi.Context(0) as cs:
            cs.synth(half, D, K, mu, L, seed, r * half, 0.9)
            st = cs.suffstat()
            fz, _ = cs.estep_posterior(*args, want_ll=False)
        tot_Fz += fz
        tot_stats = st if tot_stats is None else tuple(a + b for a, b in zip(tot_stats, st))
    np.testing.assert_allclose(tot_stats[0], Nk, rtol=1e-11)  # (two summation orders of ~N/K terms each: a few 1e-12)
    np.testing.assert_allclose(tot_stats[1], xs, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(tot_stats[2], xxs, rtol=1e-9, atol=1e-5)
    assert abs(tot_Fz - Fz) <= 1e-11 * abs(Fz)


@pytest.mark.parametrize("family", ["NormGamma", "ExpGamma"])
def test_full_size_properties_diagonal_families(family):
    """The same size-independent properties for the separable families at N = 10M, D = 64, K = 32 (bench --config
    dgmm / bemm): unit row sums, prefix rows bit-identical to a stand-alone run and equal to the oracle, shard
    additivity of the MFMA statistics and of F_z, non-increasing F."""
    N, D, K, seed = 10_000_000, 64, 32, 1006
    eg = family == "ExpGamma"
    rng = np.random.default_rng(seed)
    mu = rng.uniform(20.0, 60.0, (K, D)) if eg else rng.normal(0.0, 3.0, (K, D))
    L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
    ck = capi.C_EXPGAMMA if eg else capi.C_NORMGAMMA
    P = 4096
    with capi.Context(0) as ctx:
        ctx.synth(N, D, K, mu, L, seed, 0, 0.9)
        Xp = ctx.get_rows(0, 0, P)
        Nk, xs, xxs, Njk = ctx.suffstat_diag(second=not eg)
        assert abs(Nk.sum() - N) <= 1e-9 * N
        if not eg:
            assert np.all(xxs > 0)
        if eg:
            post = [capi.eg_mstep(1.0, Nk[k], xs[k]) for k in range(K)]
            a, w2 = np.zeros((K, D)), np.zeros((K, D))
            w1 = np.stack([-p["a"] * p["ib"] for p in post])
        else:
            post = [capi.ng_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
            a = np.stack([p["m"] for p in post])
            w2 = np.stack([-0.5 * p["nu"] / p["L"] for p in post])
            w1 = np.zeros((K, D))
        elog, _ = capi.weights_update(capi.W_DIRICHLET, Nk)
        c = (elog + np.array([p["eloglike_const"] for p in post]))[None, :]
        Fz, _ = ctx.estep_diag(a, w2, w1, c)
        qp = ctx.get_qz_rows(0, 0, P)
        qmid = ctx.get_qz_rows(0, N // 2 - 7, 1001)
        colsum = ctx.colsums()[0]
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=3, nthreads=8, ckind=ck)
        m.close()
    assert abs(colsum.sum() - N) <= 1e-9 * N
    np.testing.assert_allclose(qp.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(qmid.sum(axis=1), 1.0, rtol=1e-12)
    assert np.all(np.diff(tr) <= 1e-9 * abs(tr[0]))

    with capi.Context(0) as c2:
        c2.set_data(Xp)
        Fzp, _ = c2.estep_diag(a, w2, w1, c)
        qp2 = c2.get_qz([P])[0]
    np.testing.assert_array_equal(qp, qp2)
    logq = c + np.einsum("kd,nkd->nk", w2, (Xp[:, None, :] - a[None]) ** 2) + Xp @ w1.T
    logZ = o.logsumexp(logq)
    qref = np.exp(logq - logZ[:, None])
    big = qref > 1e-12
    assert np.max(np.abs(qp[big] - qref[big]) / qref[big]) < 1e-9
    assert abs(Fzp + logZ.sum()) <= 1e-10 * abs(logZ.sum())

    half = N // 2
    tot, tot_Fz = None, 0.0
    for r in range(2):
        with capi.Context(0) as cs:
            cs.synth(half, D, K, mu, L, seed, r * half, 0.9)
            st = cs.suffstat_diag(second=not eg)
            fz, _ = cs.estep_diag(a, w2, w1, c)
        tot_Fz += fz
        st = [x for x in st if x is not None]
        tot = st if tot is None else [p + q for p, q in zip(tot, st)]
    np.testing.assert_allclose(tot[0], Nk, rtol=1e-12)
    np.testing.assert_allclose(tot[1], xs, rtol=1e-9, atol=1e-6)
    if not eg:
        np.testing.assert_allclose(tot[2], xxs, rtol=1e-9, atol=1e-5)
    assert abs(tot_Fz - Fz) <= 1e-11 * abs(Fz)


def test_rows_beyond_32bit_element_indices():
    """N = 50M rows of D = 64 with K = 48: element indices of X (3.2e9) and of qZ (2.4e9) pass 2^31.  The same
    size-independent properties: unit row sums, the LAST rows bit-identical to a stand-alone run on just those rows,
    statistics and F_z equal to the sums over five 10M-row shards of the same Philox stream -- for the Gauss-Wishart
    kernels and for the diagonal family."""
    N, D, K, seed = 50_000_000, 64, 48, 1010
    mu, L = _mixture(D, K, seed)
    P = 2048
    with capi.Context(0) as ctx:
        ctx.synth(N, D, K, mu, L, seed, 0, 0.9)
        Xt = ctx.get_rows(0, N - P, P)
        Nk, xs, xxs, _ = ctx.suffstat()
        assert abs(Nk.sum() - N) <= 1e-9 * N
        post = [capi.gw_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
        elog, _ = capi.weights_update(capi.W_STICKBREAK, Nk)
        args = ([p["nu"] for p in post], [p["beta"] for p in post], np.stack([p["m"] for p in post]),
                np.stack([p["iW"] for p in post]), [p["logdW"] for p in post], elog[None, :])
        Fz, _ = ctx.estep_posterior(*args, want_ll=False)
        qt = ctx.get_qz_rows(0, N - P, P)
        colsum = ctx.colsums()[0]
        # diagonal family on the same rows
        dNk, dxs, dxxs, _ = ctx.suffstat_diag(second=True)
        dpost = [capi.ng_mstep(1.0, dNk[k], dxs[k], dxxs[k]) for k in range(K)]
        da = np.stack([p["m"] for p in dpost])
        dw2 = np.stack([-0.5 * p["nu"] / p["L"] for p in dpost])
        delog, _ = capi.weights_update(capi.W_DIRICHLET, dNk)
        dc = (delog + np.array([p["eloglike_const"] for p in dpost]))[None, :]
        dFz, _ = ctx.estep_diag(da, dw2, np.zeros((K, D)), dc)
        dqt = ctx.get_qz_rows(0, N - P, P)
        dcol = ctx.colsums()[0]
    assert abs(colsum.sum() - N) <= 1e-9 * N and abs(dcol.sum() - N) <= 1e-9 * N
    np.testing.assert_allclose(qt.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(dqt.sum(axis=1), 1.0, rtol=1e-12)
    with capi.Context(0) as c2:
        c2.set_data(Xt)
        c2.estep_posterior(*args, want_ll=False)
        np.testing.assert_array_equal(qt, c2.get_qz([P])[0])
        c2.estep_diag(da, dw2, np.zeros((K, D)), dc)
        np.testing.assert_array_equal(dqt, c2.get_qz([P])[0])
    S = 5
    part = N // S
    tot, dtot, tFz, tdFz = None, None, 0.0, 0.0
    for r in range(S):
        with capi.Context(0) as cs:
            cs.synth(part, D, K, mu, L, seed, r * part, 0.9)
            st = cs.suffstat()[:3]
            fz, _ = cs.estep_posterior(*args, want_ll=False)
            dst = cs.suffstat_diag(second=True)[:3]  # (of the responsibilities just computed, as above)
            dfz, _ = cs.estep_diag(da, dw2, np.zeros((K, D)), dc)
        tFz += fz
        tdFz += dfz
        tot = st if tot is None else tuple(a + b for a, b in zip(tot, st))
        dtot = dst if dtot is None else tuple(a + b for a, b in zip(dtot, dst))
    np.testing.assert_allclose(tot[0], Nk, rtol=1e-11)
    np.testing.assert_allclose(tot[1], xs, rtol=1e-9, atol=1e-5)
    np.testing.assert_allclose(tot[2], xxs, rtol=1e-9, atol=1e-4)
    np.testing.assert_allclose(dtot[0], dNk, rtol=1e-12)
    np.testing.assert_allclose(dtot[1], dxs, rtol=1e-9, atol=1e-5)
    np.testing.assert_allclose(dtot[2], dxxs, rtol=1e-9, atol=1e-4)
    assert abs(tFz - Fz) <= 1e-11 * abs(Fz) and abs(tdFz - dFz) <= 1e-11 * abs(dFz)


def test_config4_full_80M_rows_on_one_gpu():
    """BASELINE configs[3] at its FULL size on one GPU: BGMM (Dirichlet weights), N = 80M, D = 64, K = 32 -- 41 GB of X
    and 20.5 GB of qZ resident (the 8-GPU run shards exactly these rows, 10M per rank, and sums the statistics).
    Size-independent properties: unit row sums, a prefix bit-identical to a stand-alone run and equal to the oracle,
    exactly symmetric statistics, and SHARD ADDITIVITY over the eight 10M-row Philox shards the 8-GPU run would hold:
    statistics, N_k and F_z of the shards add up to the unsharded values (what the all-reduce computes); F does not
    increase over VBEM iterations with the Dirichlet weights."""
    N, D, K, seed, S = 80_000_000, 64, 32, 1004, 8
    mu, L = _mixture(D, K, seed)
    P = 4096
    with capi.Context(0) as ctx:
        ctx.synth(N, D, K, mu, L, seed, 0, 0.9)
        Xp = ctx.get_rows(0, 0, P)
        Xt = ctx.get_rows(0, N - P, P)  # rows past 2^32 elements of X
        Nk, xs, xxs, Njk = ctx.suffstat()
        assert abs(Nk.sum() - N) <= 1e-9 * N
        assert np.array_equal(xxs, np.transpose(xxs, (0, 2, 1)))
        np.testing.assert_array_equal(Njk[0], Nk)
        post = [capi.gw_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
        elog, _ = capi.weights_update(capi.W_DIRICHLET, Nk)
        args = ([p["nu"] for p in post], [p["beta"] for p in post], np.stack([p["m"] for p in post]),
                np.stack([p["iW"] for p in post]), [p["logdW"] for p in post], elog[None, :])
        Fz, _ = ctx.estep_posterior(*args, want_ll=False)
        qp = ctx.get_qz_rows(0, 0, P)
        qt = ctx.get_qz_rows(0, N - P, P)
        colsum = ctx.colsums()[0]
        Nk2, xs2, xxs2, _ = ctx.suffstat()  # statistics of the NEW responsibilities (what iteration 2 starts from)
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=2, nthreads=16)
        m.close()
    assert abs(colsum.sum() - N) <= 1e-9 * N
    np.testing.assert_allclose(qp.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(qt.sum(axis=1), 1.0, rtol=1e-12)
    assert np.all(np.diff(tr) <= 1e-9 * abs(tr[0]))

    for Xs, qs in ((Xp, qp), (Xt, qt)):  # first and last rows: bit-identical to a stand-alone run
        with capi.Context(0) as c2:
            c2.set_data(Xs)
            c2.estep_posterior(*args, want_ll=False)
            np.testing.assert_array_equal(qs, c2.get_qz([P])[0])
    cl = []
    for k in range(K):
        g = o.GaussWish(1.0, D)
        g.nu, g.beta, g.m, g.iW, g.logdW = (post[k]["nu"], post[k]["beta"], post[k]["m"], post[k]["iW"],
                                            post[k]["logdW"])
        cl.append(g)
    w = o.Dirichlet()
    w.E_logpi, w.Nk = elog, Nk
    qref, _ = o.vbexpectation(Xp, w, cl)
    big = qref > 1e-12
    assert np.max(np.abs(qp[big] - qref[big]) / qref[big]) < 1e-9

    part = N // S
    tot, tot2, tFz = None, None, 0.0
    for r in range(S):
        with capi.Context(0) as cs:
            cs.synth(part, D, K, mu, L, seed, r * part, 0.9)  # rank r's rows of the one stream
            st = cs.suffstat()[:3]
            fz, _ = cs.estep_posterior(*args, want_ll=False)
            st2 = cs.suffstat()[:3]
        tFz += fz
        tot = st if tot is None else tuple(a + b for a, b in zip(tot, st))
        tot2 = st2 if tot2 is None else tuple(a + b for a, b in zip(tot2, st2))
    np.testing.assert_allclose(tot[0], Nk, rtol=1e-11)  # (N_k is a feature of the statistics GEMM: two summation orders)
    np.testing.assert_allclose(tot[1], xs, rtol=1e-9, atol=1e-5)
    np.testing.assert_allclose(tot[2], xxs, rtol=1e-9, atol=1e-4)
    np.testing.assert_allclose(tot2[0], Nk2, rtol=1e-11)
    np.testing.assert_allclose(tot2[1], xs2, rtol=1e-9, atol=1e-5)
    np.testing.assert_allclose(tot2[2], xxs2, rtol=1e-9, atol=1e-4)
    assert abs(tFz - Fz) <= 1e-11 * abs(Fz)


def _config5_gmc(J, shards, P, iters):
    """GMC (one GDirichlet per group), J groups x 500k rows, D = 128, K = 64 of BASELINE configs[4]'s Philox stream.  The
    property set of test_full_size_properties for grouped data: per-group counts add up to the group sizes, statistics
    exactly symmetric, unit row sums, a P-row prefix of EVERY group equal to the oracle's vbexpectation with that
    group's GDirichlet weights and bit-identical to a stand-alone run, additivity over whole-group shards (the
    multi-GPU partition of SURVEY 8(e): N_jk stay local, cluster statistics and F_z add up), non-increasing F."""
    NJ, D, K, seed = 500_000, 128, 64, 1005
    mu, L = _mixture(D, K, seed)
    mix = np.stack([np.random.default_rng([seed, g]).dirichlet(np.full(K, 0.5)) for g in range(J)])
    with capi.Context(0) as ctx:
        ctx.synth_groups([NJ] * J, D, K, mu, L, seed, mix=mix, group_ids=list(range(J)))
        Xp = [ctx.get_rows(j, 0, P) for j in range(J)]
        Nk, xs, xxs, Njk = ctx.suffstat()
        assert Njk.shape == (J, K)
        np.testing.assert_allclose(Njk.sum(axis=1), NJ, rtol=1e-9)      # rows of the initial qZ sum to 1, per group
        np.testing.assert_allclose(Njk.sum(axis=0), Nk, rtol=1e-11)  # (column sums against the N_k feature of the statistics GEMM)
        assert np.array_equal(xxs, np.transpose(xxs, (0, 2, 1)))
        post = [capi.gw_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
        elog = np.stack([capi.weights_update(capi.W_GDIRICHLET, Njk[j])[0] for j in range(J)])
        args = ([p["nu"] for p in post], [p["beta"] for p in post], np.stack([p["m"] for p in post]),
                np.stack([p["iW"] for p in post]), [p["logdW"] for p in post], elog)
        Fz, _ = ctx.estep_posterior(*args, want_ll=False)
        qp = [ctx.get_qz_rows(j, 0, P) for j in range(J)]
        qlast = ctx.get_qz_rows(J - 1, NJ - P, P)                        # the very last rows of the data set
        Xlast = ctx.get_rows(J - 1, NJ - P, P)
        cols = ctx.colsums()
        F, tr, m = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=iters, nthreads=16)
        m.close()
    np.testing.assert_allclose(cols.sum(axis=1), NJ, rtol=1e-9)
    for q in qp + [qlast]:
        np.testing.assert_allclose(q.sum(axis=1), 1.0, rtol=1e-12)
    assert np.all(np.diff(tr) <= 1e-9 * abs(tr[0]))

    with capi.Context(0) as c2:  # the J x P prefix rows as a data set of their own: identical rows
        c2.set_data(Xp)
        c2.estep_posterior(*args, want_ll=False)
        for a, b in zip(qp, c2.get_qz([P] * J)):
            np.testing.assert_array_equal(a, b)
    with capi.Context(0) as c3:  # and the tail of the last group, with that group's weights
        c3.set_data(Xlast)
        c3.estep_posterior(*(args[:5] + (elog[J - 1:J],)), want_ll=False)
        np.testing.assert_array_equal(qlast, c3.get_qz([P])[0])
    cl = []
    for k in range(K):
        g = o.GaussWish(1.0, D)
        g.nu, g.beta, g.m, g.iW, g.logdW = (post[k]["nu"], post[k]["beta"], post[k]["m"], post[k]["iW"],
                                            post[k]["logdW"])
        cl.append(g)
    for j in range(J):  # against the oracle, with group j's own weights
        w = o.GDirichlet()
        w.update(Njk[j])
        np.testing.assert_allclose(w.Elogweight(), elog[j], rtol=1e-10, atol=1e-12)
        qref, _ = o.vbexpectation(Xp[j], w, cl)
        big = qref > 1e-12
        assert np.max(np.abs(qp[j][big] - qref[big]) / qref[big]) < 1e-9

    tot, tFz = None, 0.0
    for gs in shards:  # whole groups per shard
        with capi.Context(0) as cs:
            cs.synth_groups([NJ] * len(gs), D, K, mu, L, seed, mix=mix[gs], group_ids=gs)
            st = cs.suffstat()
            np.testing.assert_allclose(st[3], Njk[gs], rtol=1e-12)  # the per-group counts are local
            sub = tuple(a[gs] if i == 5 else a for i, a in enumerate(args))
            fz, _ = cs.estep_posterior(*sub, want_ll=False)
        tFz += fz
        tot = st[:3] if tot is None else tuple(a + b for a, b in zip(tot, st[:3]))
    np.testing.assert_allclose(tot[0], Nk, rtol=1e-11)
    np.testing.assert_allclose(tot[1], xs, rtol=1e-9, atol=1e-5)
    np.testing.assert_allclose(tot[2], xxs, rtol=1e-9, atol=1e-4)
    assert abs(tFz - Fz) <= 1e-11 * abs(Fz)


def test_config5_per_gpu_size_gmc():
    """BASELINE configs[4] at the size ONE of its eight GPUs holds: 8 groups x 500k rows, D = 128, K = 64; a 512-row
    prefix of every group against the oracle; three uneven whole-group shards."""
    _config5_gmc(8, ([0, 1, 2], [3, 4], [5, 6, 7]), 512, 3)


def test_config5_full_size_gmc_on_one_gpu():
    """BASELINE configs[4] at its FULL size on one GPU: GMC, J = 64 groups x 500k rows = 32M rows, D = 128, K = 64 --
    32.8 GB of X and 16.4 GB of qZ resident, 64 GDirichlet weight objects, a J x K = 4096 count block.  The shards are
    the eight 8-group blocks the 8-GPU run would hold (whole groups per GPU); a 128-row prefix of every
    group (8192 rows) goes against the oracle with that group's weights."""
    _config5_gmc(64, [list(range(8 * r, 8 * r + 8)) for r in range(8)], 128, 2)
