#!/usr/bin/env python3
"""Phase split of estep_diag_kernel at N=10M, D=64, K=32: full vs raw (no normalisation), NormGamma / ExpGamma /
general parameter sets, and K sweeps.  Usage: tools/estep_diag_probe.py [N D K]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
from libcluster_amd import capi  # noqa: E402

N, D, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 64, 32)
rng = np.random.default_rng(3)
mu = rng.normal(0, 4.0, (K, D))
L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
a = mu.copy()
w2 = -rng.uniform(0.3, 1.0, (K, D))
w1 = rng.normal(0, 0.1, (K, D))
c = rng.normal(0, 1, (1, K))
z = np.zeros((K, D))
with capi.Context(0) as ctx:
    ctx.synth(N, D, K, mu, L, 99, 0, 0.9)
    ctx.timing_enable(True)
    for name, args in (("NormGamma (w1 = 0)", (a, w2, z)), ("ExpGamma (a = w2 = 0)", (z, z, w1)),
                       ("general", (a, w2, w1))):
        for raw in (False, True):
            ctx.estep_diag(*args, c, raw=raw)
            ctx.timing_reset()
            for _ in range(5):
                ctx.estep_diag(*args, c, raw=raw)
            t = ctx.timing_get()
            ms = t["estep_ms"] / t["estep_calls"]
            print(f"{name:24s} {'raw ' if raw else 'full'} {ms:7.3f} ms  ({8.0 * N * (D + K) / ms / 1e6:6.0f} GB/s algorithmic)")
