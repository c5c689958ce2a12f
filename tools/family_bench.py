#!/usr/bin/env python3
"""Kernel times of the diagonal (NormGamma) / exponential (ExpGamma) VBEM iteration on device-resident synthetic
data, against the HBM roofline (algorithmic bytes 8 N (D + K) per kernel launch).
Usage: tools/family_bench.py N D K [NormGamma|ExpGamma] [iters]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402  (one HIP runtime)
from libcluster_amd import capi  # noqa: E402

N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fam = sys.argv[4] if len(sys.argv) > 4 else "NormGamma"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ck = capi.C_NORMGAMMA if fam == "NormGamma" else capi.C_EXPGAMMA
rng = np.random.default_rng(5)
mu = rng.normal(0, 4.0, (K, D)) if ck == capi.C_NORMGAMMA else rng.uniform(20.0, 60.0, (K, D))
L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
with capi.Context(0) as ctx:
    ctx.synth(N, D, K, mu, L, 99, 0, 0.9)
    _, _, model = ctx.vbem(capi.W_DIRICHLET, fixed_iters=1, nthreads=8, ckind=ck)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    F, tr, model = ctx.vbem(capi.W_DIRICHLET, fixed_iters=iters, nthreads=8, ckind=ck, model=model)
    dt = time.perf_counter() - t0
    kt = ctx.timing_get()
    model.close()
e = kt["estep_ms"] / kt["estep_calls"]
s = kt["suffstat_ms"] / kt["suffstat_calls"]
alg = 8.0 * N * (D + K)
print(f"{fam} N={N} D={D} K={K}: {dt / iters * 1e3:.2f} ms/iteration ({N * iters / dt / 1e6:.1f} M points/s), F={F:.6f}")
print(f"  E-step   {e:.3f} ms  -> {alg / e / 1e6:.0f} GB/s algorithmic ({alg / e / 1e6 / 8000:.2f} of 8 TB/s)")
print(f"  suffstat {s:.3f} ms  -> {alg / s / 1e6:.0f} GB/s algorithmic ({alg / s / 1e6 / 8000:.2f} of 8 TB/s)")
