#!/usr/bin/env python3
"""Wall time of the full model-selection loop (cluster(): VBEM + prune + greedy split search) on
device-resident synthetic data.  Usage: tools/learn_bench.py N D Ktrue [StickBreak|Dirichlet] [GaussWish|NormGamma|ExpGamma]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402  (one HIP runtime)
from libcluster_amd import capi  # noqa: E402

N, D, Kt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wk = capi.W_STICKBREAK if len(sys.argv) < 5 or sys.argv[4] == "StickBreak" else capi.W_DIRICHLET
fam = sys.argv[5] if len(sys.argv) > 5 else "GaussWish"
ck = {"GaussWish": capi.C_GAUSSWISH, "NormGamma": capi.C_NORMGAMMA, "ExpGamma": capi.C_EXPGAMMA}[fam]
rng = np.random.default_rng(5)
scale = float(os.environ.get("LC_LB_SCALE", "4.0"))  # spread of the cluster centres (small: overlapping clusters)
mu = rng.normal(0, scale, (Kt, D)) if fam != "ExpGamma" else rng.uniform(20.0, 60.0, (Kt, D))
if fam == "GaussWish":
    L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D)))) for _ in range(Kt)])
else:
    L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(Kt)])
with capi.Context(0) as ctx:
    ctx.synth(N, D, Kt, mu, L, 99, 0, 0.9)
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    F, model = ctx.cluster(wk, nthreads=16, ckind=ck)
    dt = time.perf_counter() - t0
    kt = ctx.timing_get()
    rounds = model.rounds()
    K = model.dims()[1]
    Ns = sorted(round(model.cluster(k)["N"]) for k in range(K))
    model.close()
its = sum(len(t) for _, t in rounds)
print(f"{fam} N={N} D={D} Ktrue={Kt}: found K={K} F={F:.6f} in {dt:.2f} s; {len(rounds)} rounds, {its} main VBEM iterations")
print(f"  E-step launches {kt['estep_calls']} ({kt['estep_ms']:.1f} ms), suff-stat launches {kt['suffstat_calls']} ({kt['suffstat_ms']:.1f} ms)")
print("  cluster sizes", Ns)
if os.environ.get("LC_LB_TRACE"):  # every round's K and free-energy trace, one JSON line (tests/test_gpu_splitsearch.py)
    import json

    print("TRACE " + json.dumps([[int(k), [float(f) for f in t]] for k, t in rounds]))
