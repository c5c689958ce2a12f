import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from libcluster_amd import capi
import bench
J, N, D, K = 8, 4_000_000, 16, 8
mu, L = bench.mixture(D, K, 77)
cfg = dict(seed=77, K=K)
with capi.Context(0) as ctx:
    rng = np.random.default_rng(3)
    mix = rng.dirichlet(np.ones(K) * 0.5, J)
    ctx.synth_groups([N // J] * J, D, K, mu, L, 77, mix=mix, group_ids=list(range(J)))
    F, tr, m = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=3, nthreads=8)
    m.close()
    ctx.timing_enable(True); ctx.timing_reset()
    F2, tr2, m = ctx.vbem(capi.W_GDIRICHLET, fixed_iters=100, nthreads=8)
    t = ctx.timing_get(); m.close()
print("F", repr(float(tr[-1])), repr(float(tr2[-1])), "fused_ms", t["fused_ms"] / max(1, t["fused_calls"]), "calls", t["fused_calls"])
