import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from libcluster_amd import capi
sys.path.insert(0, '/root/repo'); import bench
N, D, K = 10_000_000, 64, 32
mu, L = bench.mixture(D, K, 1004)
with capi.Context(0) as ctx:
    ctx.synth(N, D, K, mu, L, 1004, 0, 0.9)
    F, tr, model = ctx.vbem(capi.W_DIRICHLET, fixed_iters=2, nthreads=16)
    cl = [model.cluster(k) for k in range(K)]
    args = ([c["nu"] for c in cl], [c["beta"] for c in cl], np.stack([c["mean"] for c in cl]), np.stack([c["iW"] for c in cl]), [c["logdW"] for c in cl])
    el, _ = model.weights(0)
    ctx.timing_enable(True)
    for name, fn in (("full estep", lambda: ctx.estep_posterior(*args, el[None, :], want_ll=False)),
                     ("full estep + LL", lambda: ctx.estep_posterior(*args, el[None, :], want_ll=True)),
                     ("raw (no normalisation sweep)", lambda: capi.check(capi.lib().lc_eloglike(ctx._h, K, *[capi.dptr(np.ascontiguousarray(a, dtype=np.float64)) for a in args])))):
        ctx.timing_reset()
        for _ in range(3): fn()
        t = ctx.timing_get()
        print(f"{name:32s} {t['estep_ms']/t['estep_calls']:.3f} ms")
