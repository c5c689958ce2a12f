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
