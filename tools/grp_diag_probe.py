#!/usr/bin/env python3
"""Grouped separable families (learnDGMC / EGMC shape): kernel times of the VBEM iteration with J groups.
Usage: tools/grp_diag_probe.py [N D K J NormGamma|ExpGamma]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
from libcluster_amd import capi  # noqa: E402

N, D, K, J = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8_000_000, 64, 32, 8)
fam = sys.argv[5] if len(sys.argv) > 5 else "NormGamma"
ck = capi.C_NORMGAMMA if fam == "NormGamma" else capi.C_EXPGAMMA
rng = np.random.default_rng(5)
mu = rng.normal(0, 4.0, (K, D)) if ck == capi.C_NORMGAMMA else rng.uniform(20.0, 60.0, (K, D))
L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
with capi.Context(0) as ctx:
    mix = rng.dirichlet(np.ones(K) * 0.5, J)
    ctx.synth_groups([N // J] * J, D, K, mu, L, 77, mix=mix, group_ids=list(range(J)))
    F, tr, m = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=2, nthreads=8, ckind=ck)
    m.close()
    ctx.timing_enable(True)
    ctx.timing_reset()
    F2, tr2, m = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=20, nthreads=8, ckind=ck)
    t = ctx.timing_get()
    m.close()
print(f"{fam} N={N} D={D} K={K} J={J}: F {float(tr2[-1])!r}  E-step {t['estep_ms'] / t['estep_calls']:.3f} ms  "
      f"statistics {t['suffstat_ms'] / t['suffstat_calls']:.3f} ms")
