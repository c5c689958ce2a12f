#!/usr/bin/env python3
"""LIBCLUSTER_GPUS debugging aid: the same learner unsharded and sharded (all shards on GPU 0), results side by side."""
import os
import sys
import traceback
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import libcluster_amd as lc  # noqa: E402


def blobs(seed, n, D, K):
    rng = np.random.default_rng(seed)
    mu = rng.normal(0, 6.0, (K, D))
    z = rng.integers(0, K, n)
    return mu[z] + rng.normal(size=(n, D)) * rng.uniform(0.5, 1.2, (K, 1))[z]


learner = sys.argv[1] if len(sys.argv) > 1 else "learnGMC"
sizes = [900, 1500, 400, 1200, 700]
X = [blobs(40 + j, n, 4, 3 + (j % 2)) + (j % 2) * 3.0 for j, n in enumerate(sizes)]
for shards in (0, 2, 3, 3, 5):
    for k in ("LIBCLUSTER_GPUS", "LIBCLUSTER_GPUS_SAME_DEVICE"):
        os.environ.pop(k, None)
    if shards:
        os.environ["LIBCLUSTER_GPUS"] = str(shards)
        os.environ["LIBCLUSTER_GPUS_SAME_DEVICE"] = "1"
    try:
        F, q, w, mu, cov, info = getattr(lc, learner)(X, return_info=True)
        print(f"shards={shards}: F={F:.9f} K={info['K']} rounds={[(k, len(t)) for k, t in info['rounds']]}", flush=True)
    except Exception:  # noqa: BLE001
        print(f"shards={shards}: FAILED", flush=True)
        traceback.print_exc()
