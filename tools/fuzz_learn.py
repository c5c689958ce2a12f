#!/usr/bin/env python3
"""Randomised parity sweep of the full learners (model selection: VBEM + prune + greedy split search) against the
oracle: every round's K and free energies, final K and F.  Usage: tools/fuzz_learn.py [cases] [seed]
LC_FUZZ_CACHE=1: only the shapes whose model selection runs on cached distances and moved-row statistics (Gauss-Wishart
learners, D > 16, dense), from well separated to heavily overlapping mixtures."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import torch  # noqa: F401,E402
import lc_oracle as o  # noqa: E402
import libcluster_amd as lc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
FLAT = ["learnVDP", "learnBGMM", "learnDGMM", "learnBEMM"]
GROUPED = ["learnGMC", "learnSGMC", "learnDGMC", "learnEGMC"]
CACHE_ONLY = os.environ.get("LC_FUZZ_CACHE") == "1"
if CACHE_ONLY:
    FLAT, GROUPED = ["learnVDP", "learnBGMM"], ["learnGMC", "learnGMC", "learnSGMC"]
fails, t0 = [], time.time()
for case in range(cases):
    kind = rng.choice(["flat", "grouped", "topic"], p=[0.45, 0.4, 0.15])
    D = int(rng.choice([1, 2, 3, 5, 8, 17, 23, 40, 70, 130], p=[0.15, 0.15, 0.15, 0.15, 0.12, 0.12, 0.06, 0.05, 0.03, 0.02]))
    Kt = int(rng.integers(1, 7))
    cent = rng.normal(0, 6.0, (Kt, D))
    spread = rng.uniform(0.4, 1.5)
    if CACHE_ONLY:
        kind = rng.choice(["flat", "grouped"], p=[0.6, 0.4])
        D = int(rng.choice([17, 20, 23, 33, 40, 70, 130, 200], p=[0.16, 0.16, 0.16, 0.16, 0.14, 0.12, 0.06, 0.04]))
        Kt = int(rng.integers(2, 9))
        cent = rng.normal(0, float(rng.choice([6.0, 2.0, 0.7, 0.35])), (Kt, D))

    def draw(n, positive):
        z = rng.integers(0, Kt, n)
        x = cent[z] + rng.normal(size=(n, D)) * spread
        return np.abs(x) + 0.05 if positive else x

    tag, err = None, None
    try:
        if kind == "flat":
            name = str(rng.choice(FLAT))
            X = draw(int(rng.integers(30, 1500)), name == "learnBEMM")
            prior = float(rng.choice([1.0, 0.3, 3.0]))
            maxc = int(rng.choice([-1, -1, 2, 5]))
            tag = f"case {case}: {name} N={X.shape[0]} D={D} Ktrue={Kt} prior={prior} maxclusters={maxc}"
            tr = []
            # ("Free energy increase!" inside a split candidate's vbem is reference behaviour, cluster.cpp:229-230: both
            # sides must then raise)
            try:
                Fo, _, _, clo = getattr(o, name)(X, float(np.float32(prior)), maxc, trace=tr)
                oerr = None
            except (RuntimeError, ValueError, FloatingPointError) as e:
                oerr = e
            try:
                res = getattr(lc, name)(X, prior=prior, maxclusters=maxc, threads=2, return_info=True)
                gerr = None
            except (RuntimeError, ValueError, ArithmeticError) as e:
                gerr = e
            if oerr is not None or gerr is not None:
                ok = (oerr is not None) == (gerr is not None)
                if not ok:
                    err = f"oracle: {oerr!r}; gpu: {gerr!r}"
            else:
                F, info = res[0], res[-1]
                ok = info["K"] == len(clo) and [k for k, _ in info["rounds"]] == [k for k, _ in tr]
                ok = ok and all(np.allclose(a, b, rtol=1e-7) for (_, a), (_, b) in zip(info["rounds"], tr))
                ok = ok and abs(F - Fo) <= 1e-8 * abs(Fo)
        elif kind == "grouped":
            name = str(rng.choice(GROUPED))
            J = int(rng.integers(2, 7))
            X = [draw(int(rng.integers(1, 500)), name == "learnEGMC") for _ in range(J)]
            sparse = bool(rng.integers(0, 2))
            tag = f"case {case}: {name} J={J} N={[x.shape[0] for x in X]} D={D} Ktrue={Kt} sparse={sparse}"
            tr = []
            try:
                Fo, _, _, clo = getattr(o, name)(X, 1.0, -1, sparse, trace=tr)
                oerr = None
            except (RuntimeError, ValueError, FloatingPointError) as e:
                oerr = e
            try:
                res = getattr(lc, name)(X, sparse=sparse, threads=2, return_info=True)
                gerr = None
            except (RuntimeError, ValueError, ArithmeticError) as e:
                gerr = e
            if oerr is not None or gerr is not None:
                ok = (oerr is not None) == (gerr is not None) or "zero-size" in str(oerr)
                if not ok:
                    err = f"oracle: {oerr!r}; gpu: {gerr!r}"
            else:
                F, info = res[0], res[-1]
                ok = info["K"] == len(clo) and [k for k, _ in info["rounds"]] == [k for k, _ in tr]
                ok = ok and all(np.allclose(a, b, rtol=1e-7) for (_, a), (_, b) in zip(info["rounds"], tr))
                ok = ok and abs(F - Fo) <= 1e-8 * abs(Fo)
        else:
            mcm = bool(rng.integers(0, 2))
            J, I, maxT = int(rng.integers(1, 4)), int(rng.integers(2, 7)), int(rng.integers(1, 5))
            X = [[draw(int(rng.integers(1, 200)), False) for _ in range(I)] for _ in range(J)]
            W = [rng.normal(size=(I, 2)) * 3 for _ in range(J)] if mcm else None
            maxT = min(maxT, J * I)
            qY0 = [o.random_qY(I, maxT, rng) for _ in range(J)]
            tag = f"case {case}: {'learnMCM' if mcm else 'learnSCM'} J={J} I={I} maxT={maxT} D={D} Ktrue={Kt}"
            tr = []
            if mcm:
                Fo, _, _, _, wto, _, clo = o.learnMCM(W, X, maxT=maxT, qY0=qY0, trace=tr)
                res = lc.learnMCM(W, X, trunc=maxT, qY0=qY0, threads=2, return_info=True)
            else:
                Fo, _, _, _, wto, clo = o.learnSCM(X, maxT=maxT, qY0=qY0, trace=tr)
                res = lc.learnSCM(X, trunc=maxT, qY0=qY0, threads=2, return_info=True)
            F, info = res[0], res[-1]
            ok = (info["T"], info["K"]) == (len(wto), len(clo))
            ok = ok and [(t, k) for t, k, _ in info["rounds"]] == [(t, k) for t, k, _ in tr]
            ok = ok and all(np.allclose(a, b, rtol=1e-7) for (_, _, a), (_, _, b) in zip(info["rounds"], tr))
            ok = ok and abs(F - Fo) <= 1e-8 * abs(Fo)
        if os.environ.get("LC_FUZZ_VERBOSE"):
            print(tag, "ok" if ok else "MISMATCH", flush=True)
        if not ok:
            fails.append(tag + (f" -> {err}" if err else " -> trace / K / F mismatch"))
    except Exception as e:  # noqa: BLE001
        fails.append(f"{tag} -> {type(e).__name__}: {e}")
print(f"{cases} cases in {time.time() - t0:.0f} s, {len(fails)} failures")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
