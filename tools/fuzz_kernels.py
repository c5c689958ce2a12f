#!/usr/bin/env python3
"""Kernel-level randomised checks at mid sizes (chunk / slice / row-group / batch boundaries): the statistics of all
three families against numpy, and the E-step of all three families on a random subset of rows against the oracle
(the E-step is row-independent given the posterior).  Usage: tools/fuzz_kernels.py [cases] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import torch  # noqa: F401,E402
import lc_oracle as o  # noqa: E402
from libcluster_amd import capi  # noqa: E402
from scipy.special import digamma  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails, t0 = [], time.time()
for case in range(cases):
    D = int(rng.choice([1, 3, 8, 16, 17, 23, 27, 32, 33, 42, 47, 53, 64, 65, 85, 96, 128, 130, 192, 300]))
    K = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 32, 33, 47, 64, 65, 100]))
    J = int(rng.choice([1, 1, 2, 3, 7, 31]))
    Ntot = int(rng.integers(1, max(2, int(3e8 / (K * D * D + 2000)))))
    Ntot = min(Ntot, 400_000)
    cuts = np.sort(rng.integers(0, Ntot + 1, J - 1)) if J > 1 else np.array([], dtype=int)
    sizes = np.diff(np.concatenate([[0], cuts, [Ntot]])).astype(int).tolist()
    fam = int(rng.integers(0, 3))
    X = [rng.normal(size=(n, D)) * 1.5 + rng.normal(size=(1, D)) for n in sizes]
    if fam == 2:
        X = [np.abs(x) + 0.01 for x in X]
    q0 = [rng.dirichlet(np.ones(K) * 0.5, n) if n else np.zeros((0, K)) for n in sizes]
    tag = f"case {case}: fam={fam} D={D} K={K} J={J} N={Ntot}"
    try:
        Xa, qa = np.vstack(X), np.vstack(q0)
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            if fam == 0:
                Nk, xs, xxs, Njk = ctx.suffstat()
                ref2 = np.einsum("nk,nd,ne->kde", qa, Xa, Xa, optimize=True) if Ntot * K * D * D < 4e9 else None
            else:
                Nk, xs, xxs, Njk = ctx.suffstat_diag(second=fam == 1)
                ref2 = qa.T @ (Xa * Xa) if fam == 1 else None
            ok = np.allclose(Nk, qa.sum(axis=0), rtol=1e-10, atol=1e-12) and np.allclose(xs, qa.T @ Xa, rtol=1e-9, atol=1e-8)
            if ref2 is not None:
                ok = ok and np.allclose(xxs, ref2, rtol=1e-9, atol=1e-7)
            ok = ok and np.allclose(Njk, np.stack([q.sum(axis=0) for q in q0]), rtol=1e-10, atol=1e-12)
            # posterior from these statistics, E-step, compare a subset of rows with the oracle
            cf = [o.GaussWish, o.NormGamma, o.ExpGamma][fam]
            cl = [cf(1.0, D) for _ in range(K)]
            for k, c in enumerate(cl):
                if fam == 0:
                    c.addstats(Nk[k], xs[k], xxs[k])
                elif fam == 1:
                    c.addstats(Nk[k], xs[k], xxs[k])
                else:
                    c.addstats(Nk[k], xs[k], None)
                c.update()
            w = [o.GDirichlet() if J > 1 else o.Dirichlet() for _ in range(J)]
            for j in range(J):
                w[j].update(Njk[j])
            el = np.stack([x.Elogweight() for x in w])
            if fam == 0:
                Fz, _ = ctx.estep_posterior([c.nu for c in cl], [c.beta for c in cl], np.stack([c.m for c in cl]),
                                            np.stack([c.iW for c in cl]), [c.logdW for c in cl], el, want_ll=False)
            else:
                a, w2, w1, cst = np.zeros((K, D)), np.zeros((K, D)), np.zeros((K, D)), np.zeros(K)
                for k, c in enumerate(cl):
                    if fam == 1:
                        a[k], w2[k] = c.m, -0.5 * c.nu / c.L
                        cst[k] = 0.5 * (D * (digamma(c.nu) - np.log(2 * np.pi) - 1.0 / c.beta) - c.logL)
                    else:
                        w1[k] = -c.a * c.ib
                        cst[k] = D * digamma(c.a) - c.logb
                Fz, _ = ctx.estep_diag(a, w2, w1, el + cst[None, :])
            q = ctx.get_qz(sizes)
        Fzref = 0.0
        for j in range(J):
            if sizes[j] == 0:
                continue
            sub = rng.choice(sizes[j], size=min(sizes[j], 300), replace=False)
            qr, _ = o.vbexpectation(X[j][sub], w[j], cl)
            big = qr > 1e-12
            if big.any():
                ok = ok and float(np.max(np.abs(q[j][sub][big] - qr[big]) / qr[big])) < 1e-7
            ok = ok and np.allclose(q[j].sum(axis=1), 1.0, rtol=1e-10)
        if Ntot * K * D * (D if fam == 0 else 1) < 2e8:  # full F_z when the oracle is cheap enough
            for j in range(J):
                if sizes[j]:
                    Fzref += o.vbexpectation(X[j], w[j], cl)[1]
            ok = ok and abs(Fz - Fzref) <= 1e-9 * max(1.0, abs(Fzref))
        if not ok:
            fails.append(tag)
    except Exception as e:  # noqa: BLE001
        fails.append(f"{tag} -> {type(e).__name__}: {e}")
print(f"{cases} cases in {time.time() - t0:.0f} s, {len(fails)} failures")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
