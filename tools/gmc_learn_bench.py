#!/usr/bin/env python3
"""Wall time of learnGMC's model-selection loop (cluster() with GDirichlet weights over J groups) on device-resident
synthetic groups whose mixing proportions differ.  Usage: tools/gmc_learn_bench.py J rows_per_group D Ktrue"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402  (one HIP runtime)
from libcluster_amd import capi  # noqa: E402

J, n, D, Kt = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(9)
mu = rng.normal(0, 4.0, (Kt, D))
L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D)))) for _ in range(Kt)])
mix = rng.dirichlet(np.full(Kt, 0.5), J)
with capi.Context(0) as ctx:
    ctx.synth_groups([n] * J, D, Kt, mu, L, 77, mix=mix)
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    F, model = ctx.cluster(capi.W_GDIRICHLET, nthreads=16)
    dt = time.perf_counter() - t0
    kt = ctx.timing_get()
    rounds = model.rounds()
    K = model.dims()[1]
    model.close()
print(f"GMC J={J} x {n} rows D={D} Ktrue={Kt}: found K={K} F={F:.6f} in {dt:.2f} s; {len(rounds)} rounds, "
      f"{sum(len(t) for _, t in rounds)} main VBEM iterations")
print(f"  E-step launches {kt['estep_calls']} ({kt['estep_ms']:.1f} ms), suff-stat launches {kt['suffstat_calls']} ({kt['suffstat_ms']:.1f} ms)")
