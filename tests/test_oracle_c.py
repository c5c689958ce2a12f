"""The C port (bench.py's CPU baseline) against the numpy oracle."""
import numpy as np
import pytest

import lc_oracle as o
import lc_oracle_c as oc


@pytest.mark.parametrize("N,D,K,nt", [(257, 2, 3, 1), (1000, 16, 8, 2), (333, 64, 5, 3), (100, 128, 2, 1)])
def test_c_port_matches_numpy_oracle(N, D, K, nt):
    rng = np.random.default_rng(N + D)
    X = rng.normal(size=(N, D)) * 1.3 + rng.integers(0, K, (N, 1))
    q0 = rng.dirichlet(np.ones(K) * 0.4, N)
    w = o.Dirichlet()
    cl = [o.GaussWish(1.0, D) for _ in range(K)]
    w.update(o.updateSS(X, q0, cl))
    Nk, xs, xxs = oc.suffstat(X, q0, nt)
    np.testing.assert_allclose(Nk, [c.N_s for c in cl], rtol=1e-12)
    np.testing.assert_allclose(xs, np.stack([c.x_s for c in cl]), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(xxs, np.stack([c.xx_s for c in cl]), rtol=1e-10, atol=1e-9)
    for c in cl:
        c.update()
    qref, Fzref = o.vbexpectation(X, w, cl)
    q, Fz = oc.estep(X, [c.nu for c in cl], [c.beta for c in cl], np.stack([c.m for c in cl]),
                     np.stack([c.iW for c in cl]), [c.logdW for c in cl], w.Elogweight(), nt)
    assert abs(Fz - Fzref) < 1e-10 * abs(Fzref)
    np.testing.assert_allclose(q, qref, rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize("wf", [o.Dirichlet, o.StickBreak])
def test_c_port_fixed_k_vbem_matches_numpy_oracle(wf):
    """bench.py's parity leg runs the C port's VBEM (C data passes + the numpy oracle's M-step and free energy) on
    1e6 rows; here the same loop against lc_oracle.vbem_fixed on an OVERLAPPING mixture (soft responsibilities),
    including a row count that is not a multiple of the C tile sizes."""
    rng = np.random.default_rng(11)
    N, D, K = 2309, 12, 6
    mu = rng.normal(0, 0.6, (K, D))
    X = mu[rng.integers(0, K, N)] + rng.normal(size=(N, D))
    q0 = rng.dirichlet(np.ones(K), N)
    Fref, _, qref, _, _ = o.vbem_fixed([X], [q0], wf, 1.0, 3)
    F, q = oc.vbem_fixed(X, q0, wf, 1.0, 3, nthreads=3)
    np.testing.assert_allclose(F, Fref, rtol=1e-12)
    soft = (qref[0] > 1e-6) & (qref[0] < 1 - 1e-6)
    assert soft.mean() > 0.5  # the comparison is not vacuous
    np.testing.assert_allclose(q, qref[0], rtol=1e-9, atol=1e-14)
