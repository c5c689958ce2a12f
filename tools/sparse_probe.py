#!/usr/bin/env python3
"""What the `sparse` option and the exact zero-skipping buy on grouped data where every group uses a few clusters.
Usage: tools/sparse_probe.py [rows_per_group D K J]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from libcluster_amd import capi  # noqa: E402

n, D, K, J = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (500_000, 64, 32, 8)
mu, L = bench.mixture(D, K, 1005)
mix = np.stack([np.random.default_rng([1005, g]).dirichlet(np.full(K, float(sys.argv[5]) if len(sys.argv) > 5 else 0.1)) for g in range(J)])
print("clusters with > 1 % of a group's mass:", (mix > 0.01).sum(axis=1).tolist())
for label, sparse, skip in (("dense", False, False), ("dense + zero skipping", False, True), ("sparse", True, False)):
    with capi.Context(0) as ctx:
        ctx.synth_groups([n] * J, D, K, mu, L, 1005, mix=mix, group_ids=list(range(J)))
        ctx.set_skip_zero(skip)
        _, _, model = ctx.vbem(capi.W_GDIRICHLET, sparse=sparse, fixed_iters=4, nthreads=8)
        ctx.timing_enable(True)
        ctx.timing_reset()
        F, tr, model = ctx.vbem(capi.W_GDIRICHLET, sparse=sparse, fixed_iters=6, nthreads=8, model=model)
        t = ctx.timing_get()
        model.close()
    print(f"{label:24s} E-step {t['estep_ms'] / t['estep_calls']:7.3f} ms  suff-stats "
          f"{t['suffstat_ms'] / t['suffstat_calls']:7.3f} ms  F = {F:.6f}")
