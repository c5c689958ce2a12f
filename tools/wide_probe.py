#!/usr/bin/env python3
"""Gauss-Wishart kernels at any width (host-generated data; the in-between layouts of 48 / 96 columns and
observations wider than 128 columns): time of the E-step and of the statistics pass against the fp64 MFMA peak.  Usage: tools/wide_probe.py [N D K]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
from libcluster_amd import capi  # noqa: E402

N, D, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (400_000, 256, 16)
rng = np.random.default_rng(1)
X = rng.normal(size=(N, D)) + rng.integers(0, K, (N, 1))
q = rng.dirichlet(np.ones(K) * 0.5, N)
nu = np.full(K, D + 2.0)
beta = np.ones(K)
m = rng.normal(size=(K, D))
iW = np.stack([np.eye(D) * (1.0 + 0.1 * k) for k in range(K)])
logdW = np.array([-np.linalg.slogdet(w)[1] for w in iW])
elw = np.log(np.full((1, K), 1.0 / K))
DP = (D + 63) // 64 * 64 if D > 128 else max(16, (D + 15) // 16 * 16)  # lck::padded_dim_wide
with capi.Context(0) as ctx:
    ctx.set_data(X)
    ctx.set_qz(q)
    ctx.timing_enable(True)
    for rep in range(3):
        ctx.timing_reset()
        ctx.suffstat()
        ctx.estep_posterior(nu, beta, m, iW, logdW, elw)
        t = ctx.timing_get()
    es, ss = t["estep_ms"] / t["estep_calls"], t["suffstat_ms"] / t["suffstat_calls"]
    flop = 2.0 * N * K * (DP * (DP + 4) / 2)
    print(f"N={N} D={D} (DP={DP}) K={K}: E-step {es:8.3f} ms ({flop / es / 1e9:5.1f} TFLOP/s)   "
          f"statistics {ss:8.3f} ms ({flop / ss / 1e9:5.1f} TFLOP/s)   of 78.6 peak")
