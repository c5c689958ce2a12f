#!/usr/bin/env python3
"""Wall time of learnSCM on many documents (host document loops vs device passes).
Usage: tools/topic_bench.py [docs rows_per_doc D K T threads]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
import libcluster_amd as lc  # noqa: E402

I, n, D, K, T, thr = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (20000, 100, 8, 10, 4, 16)
rng = np.random.default_rng(2)
mu = rng.normal(0, 6.0, (K, D))
mix = rng.dirichlet(np.full(K, 0.3), T)
X = []
for i in range(I):
    z = rng.choice(K, size=n, p=mix[rng.integers(0, T)])
    X.append(mu[z] + rng.normal(size=(n, D)))
qY0 = [np.abs(rng.uniform(-1, 1, (I, 2 * T)))]
qY0[0] /= qY0[0].sum(axis=1, keepdims=True)
for threads in (1, thr):
    t0 = time.perf_counter()
    f, qY, qZ, wi, ws, m, c, info = lc.learnSCM([X], trunc=2 * T, qY0=qY0, threads=threads, return_info=True)
    dt = time.perf_counter() - t0
    its = sum(len(r[2]) for r in info["rounds"])
    print(f"docs={I} rows={I * n} D={D}: threads={threads}: T={info['T']} K={info['K']} F={f:.4f} in {dt:.2f} s "
          f"({len(info['rounds'])} rounds, {its} VBEM iterations)")
