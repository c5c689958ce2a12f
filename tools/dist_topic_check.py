#!/usr/bin/env python3
"""learnSCM / learnMCM with whole groups sharded over ranks, one process per rank.

    python tools/dist_topic_check.py [scm|mcm]                      # single rank
    LC_DIST_BACKEND=gloo LC_ALL_RANKS_ON_GPU0=1 python -m torch.distributed.run --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port P tools/dist_topic_check.py [scm|mcm]

Every world size sees the same 4 groups x 6 documents and the same initial qY; rank r holds groups r, r+W, ...
Rank 0 prints one JSON line (F, T, K, rounds)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from libcluster_amd import capi  # noqa: E402
from libcluster_amd import dist as lcd  # noqa: E402

mcm = len(sys.argv) > 1 and sys.argv[1] == "mcm"
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dev = 0 if os.environ.get("LC_ALL_RANKS_ON_GPU0") else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist

    backend = os.environ.get("LC_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
rng = np.random.default_rng(4)
J, I, n, D, K, T, maxT = 4, 6, 400, 4, 5, 2, 4
mu = rng.normal(0, 6.0, (K, D))
mix = rng.dirichlet(np.full(K, 0.4), T)
mw = rng.normal(0, 4.0, (T, 2))
X, W, qY0 = [], [], []
for j in range(J):
    Xj, Wj = [], []
    for i in range(I):
        t = rng.integers(0, T)
        z = rng.choice(K, size=n, p=mix[t])
        Xj.append(mu[z] + rng.normal(size=(n, D)))
        Wj.append(mw[t] + 0.7 * rng.normal(size=2))
    X.append(Xj)
    W.append(np.array(Wj))
    r = np.abs(rng.uniform(-1, 1, (I, maxT)))
    qY0.append(r / r.sum(axis=1, keepdims=True))
mine = list(range(rank, J, world))
hook = lcd.make_device_hook(dev) if world > 1 else None
F, m = capi.learn_topic([X[j] for j in mine], [W[j] for j in mine] if mcm else None, [qY0[j] for j in mine], 1.0, 1.0,
                        maxT, -1, False, 2, dev, allreduce=hook, stream=torch.cuda.current_stream().cuda_stream)
d = m.dims()
out = {"world": world, "F": F, "T": d["T"], "K": d["K"], "rounds": m.rounds(),
       "N": [m.cluster(0, k)["N"] for k in range(d["K"])]}
m.close()
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
