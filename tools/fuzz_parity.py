#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the oracle: random shapes (N, D, K, J, ragged and empty groups),
weight kinds, cluster families, sparse on/off, 1-3 fixed VBEM iterations each.  Prints failures and a summary.
Usage: tools/fuzz_parity.py [cases] [seed]"""
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
from libcluster_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
WK = [(o.Dirichlet, capi.W_DIRICHLET), (o.StickBreak, capi.W_STICKBREAK), (o.GDirichlet, capi.W_GDIRICHLET)]
CK = [(o.GaussWish, capi.C_GAUSSWISH), (o.NormGamma, capi.C_NORMGAMMA), (o.ExpGamma, capi.C_EXPGAMMA)]
fails, t0 = [], time.time()
skipped_ties = 0
for case in range(cases):
    D = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 23, 31, 32, 33, 48, 64, 65, 100, 128, 129, 160, 257]))
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 40, 65]))
    J = int(rng.choice([1, 1, 1, 2, 3, 5, 9, 40]))
    budget = max(40, int(200000 / (D * D * K) * 40))  # keep the numpy oracle fast
    sizes = [int(rng.integers(0, budget)) for _ in range(J)]
    if sum(sizes) < K + 2:
        sizes[0] += K + 5
    wf, wk = WK[int(rng.integers(0, 3))]
    cf, ck = CK[int(rng.integers(0, 3))]
    sparse = bool(rng.integers(0, 2)) and J > 1
    iters = int(rng.integers(1, 4))
    cent = rng.normal(0, 4.0, (K, D))
    # a quarter of the cases sit 10 or 100 sigma away from the origin (y = A x - b cancels like eps * offset / sigma;
    # beyond ~1e3 the reference's own S_k - N_k xbar xbar^T moves by more than this sweep's tolerance: tests/test_gpu_offset.py)
    offset = float(rng.choice([0.0, 0.0, 0.0, 10.0, 100.0])) * rng.uniform(0.5, 1.5, D)
    cent = cent + offset
    X, q0 = [], []
    for n in sizes:
        z = rng.integers(0, K, n)
        x = cent[z] + rng.normal(size=(n, D)) * rng.uniform(0.3, 2.0)
        if cf is o.ExpGamma:
            x = np.abs(x) + 0.01
        X.append(x)
        hard = rng.random() < 0.5
        q = np.zeros((n, K))
        if hard and n:
            q[np.arange(n), rng.integers(0, min(K, 3), n) if sparse else z] = 1.0
        else:
            q = rng.dirichlet(np.ones(K) * 0.4, n) if n else q
        q0.append(q)
    tag = f"case {case}: offset={offset.max():.0f} D={D} K={K} J={J} N={sizes if J < 6 else sum(sizes)} {wf.__name__}/{cf.__name__} sparse={sparse} it={iters}"
    if os.environ.get("LC_FUZZ_VERBOSE"):
        print(tag, flush=True)
    if os.environ.get("LC_FUZZ_ONLY") and case != int(os.environ["LC_FUZZ_ONLY"]):
        continue
    try:
        tro, _, qo, wo, _ = o.vbem_fixed(X, q0, wf, 1.0, iters, sparse, cf)
    except Exception as e:  # the oracle itself rejects the case (e.g. non-PD): the GPU path must fail too
        if "zero-size array" in str(e):
            continue  # sparse mode, a group without any active cluster: logsumexp over zero columns is undefined
                      # behaviour in the reference (Eigen maxCoeff of an empty row); nothing to compare
        try:
            with capi.Context(0) as ctx:
                ctx.set_data(X)
                ctx.set_qz(q0)
                ctx.vbem(wk, fixed_iters=iters, sparse=sparse, ckind=ck)
            fails.append(tag + f" -> oracle raised {type(e).__name__} ({e}) but the GPU path did not")
        except Exception:
            pass
        continue
    try:
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            F, tr, model = ctx.vbem(wk, fixed_iters=iters, sparse=sparse, ckind=ck, nthreads=2)
            q = ctx.get_qz(sizes)
            model.close()
    except Exception as e:
        if not np.all(np.isfinite(tro)):
            continue  # the oracle's trace is not finite either (e.g. a group whose clusters are all inactive)
        fails.append(tag + f" -> GPU path raised {type(e).__name__}: {e}")
        continue
    ok = np.allclose(tr, tro, rtol=1e-8, atol=0)
    if not (np.all(np.isfinite(tro)) and np.all(np.isfinite(tr))):
        ok = np.array_equal(np.isfinite(tr), np.isfinite(tro))
    dq = 0.0
    for a, b in zip(q, qo):
        if a.size:
            big = b > 1e-10
            if big.any():
                dq = max(dq, float(np.max(np.abs(a[big] - b[big]) / b[big])))
            dq = max(dq, float(np.max(np.abs(a - b))) if not np.isnan(b).any() else 0.0)
    if ok and not dq < 1e-6 and wf is not o.Dirichlet:
        # The stick-breaking weights sort the clusters by their counts (distributions.cpp:157-164).  Counts that differ
        # only in their last bits (hard assignments: 1 + 2.5e-10 against 1 + 1.6e-48) are ordered by summation noise, and
        # E[log pi] of the clusters involved then differs by O(1) between ANY two implementations while F does not
        # (seed 101 case 443).  Such a case says nothing about the kernels: F decides it.
        def near_tie(nk):
            v = np.sort(np.asarray(nk, dtype=float))[::-1]
            a, b = v[:-1], v[1:]
            # (exactly equal non-zero counts included: the two sides' exponentials differ by an ulp, so counts that
            # round to the same double in the oracle need not on the device -- seed 41 cases 972 and 2183)
            return bool(np.any((np.abs(a - b) <= 1e-9 * np.maximum(np.abs(a), 1e-300)) & (a > 1e-12)))
        if any(near_tie(w.Nk) for w in wo):
            skipped_ties += 1
            continue
    if (not ok or not dq < 1e-6) and wf is not o.Dirichlet and iters > 1 and np.all(np.isfinite(tro)):
        # ... and a tie in an EARLIER iteration reaches F one iteration later (the two sides then run on different E[log pi]
        # from there on: seed 202 case 465, hard assignments, counts 4, 4, 3, 3, 3, 2, ...).  The traces have to agree up to
        # and including the first iteration whose counts tie; what follows says nothing about the kernels.
        def near_tie2(nk):
            v = np.sort(np.asarray(nk, dtype=float))[::-1]
            a, b = v[:-1], v[1:]
            return bool(np.any((np.abs(a - b) <= 1e-9 * np.maximum(np.abs(a), 1e-300)) & (a > 1e-12)))
        tied_at = None
        for t in range(1, iters):
            _, _, _, wt, _ = o.vbem_fixed(X, q0, wf, 1.0, t, sparse, cf)
            if any(near_tie2(w.Nk) for w in wt):
                tied_at = t
                break
        if tied_at is not None and np.allclose(tr[:tied_at], tro[:tied_at], rtol=1e-8, atol=0):
            skipped_ties += 1
            continue
    if not ok or not dq < 1e-6:
        if os.environ.get("LC_FUZZ_ONLY"):
            np.savez("gpurun_out/fuzz_case.npz", **{f"X{g}": x for g, x in enumerate(X)}, **{f"q0{g}": x for g, x in enumerate(q0)},
                     **{f"q{g}": x for g, x in enumerate(q)}, **{f"qo{g}": x for g, x in enumerate(qo)})
        where = ""
        for g, (a, b) in enumerate(zip(q, qo)):  # the worst entry, for the report
            if a.size and np.max(np.abs(a - b)) > 1e-9:
                r, k = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
                where += f" [group {g} row {r} k {k}: gpu {a[r, k]:.6e} oracle {b[r, k]:.6e}; row sums {a[r].sum():.6f} {b[r].sum():.6f}]"
        fails.append(tag + f" -> F {tr} vs {tro}, dq={dq:.3e}" + where[:600])
print(f"{cases} cases in {time.time() - t0:.0f} s, {len(fails)} failures" + (f" ({skipped_ties} decided by F alone: near-tied counts in the stick-breaking order)" if skipped_ties else ""))
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
