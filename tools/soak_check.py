#!/usr/bin/env python3
"""Soak check: a few hundred learner calls of mixed shapes in one process; device memory (block cache plateau) and host
RSS must stay flat.  Usage: tools/soak_check.py"""
import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import libcluster_amd as lc
from libcluster_amd import capi
rng = np.random.default_rng(0)
free0, tot = torch.cuda.mem_get_info()
import resource
for it in range(300):
    D = int(rng.choice([2, 5, 17, 40, 70, 130]))
    N = int(rng.integers(50, 3000))
    X = rng.normal(size=(N, D)) + rng.integers(0, 3, (N, 1)) * 4
    fn = [lc.learnBGMM, lc.learnVDP, lc.learnDGMM][it % 3]
    fn(X)
    if it % 3 == 0:
        Xg = [X[: N // 2], X[N // 2:]]
        lc.learnGMC(Xg, sparse=bool(it % 2))
    if it % 50 == 49:
        free, _ = torch.cuda.mem_get_info()
        print(it + 1, "device MB used since start:", (free0 - free) / 2**20, "host maxrss MB:", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, flush=True)
capi.trim_cache()
free, _ = torch.cuda.mem_get_info()
print("after trim: device MB used since start:", (free0 - free) / 2**20)
