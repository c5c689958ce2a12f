#!/usr/bin/env python3
"""Model selection (lc_cluster) on row-sharded synthetic data, one process per rank.

    python tools/dist_cluster_check.py N D K                      # single rank
    LC_DIST_BACKEND=gloo LC_ALL_RANKS_ON_GPU0=1 python -m torch.distributed.run --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port P tools/dist_cluster_check.py N D K

Rank r generates rows [r*N/W, (r+1)*N/W) of the same Philox stream, so every world size sees the
same data set; rank 0 prints one JSON line (F, K, rounds).  With nccl (default) ranks use their own
GPU; gloo + LC_ALL_RANKS_ON_GPU0 exercises the multi-rank path on a 1-GPU box."""
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

N, D, Kt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
MODE = sys.argv[4] if len(sys.argv) > 4 else "rows"  # "rows": BGMM, row blocks; "groups": GMC, whole groups per rank
FAMILY = sys.argv[5] if len(sys.argv) > 5 else "GaussWish"  # cluster family: GaussWish | NormGamma | ExpGamma
CK = {"GaussWish": capi.C_GAUSSWISH, "NormGamma": capi.C_NORMGAMMA, "ExpGamma": capi.C_EXPGAMMA}[FAMILY]
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
rng = np.random.default_rng(3)
mu = rng.normal(0, 5.0, (Kt, D))
L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D)))) for _ in range(Kt)])
if FAMILY != "GaussWish":  # axis-aligned components; the exponential family needs x >= 0
    L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(Kt)])
    if FAMILY == "ExpGamma":  # components that differ in magnitude (what an exponential mixture can tell apart)
        scale = 9.0 ** np.arange(Kt)
        mu = scale[:, None] * rng.uniform(0.8, 1.2, (Kt, D))
        L = np.stack([np.diag(0.1 * scale[k] * rng.uniform(0.5, 1.5, D)) for k in range(Kt)])
with capi.Context(dev, torch.cuda.current_stream().cuda_stream) as ctx:
    if MODE == "rows":
        lo, hi = lcd.shard_rows(N, world, rank)
        ctx.synth(hi - lo, D, Kt, mu, L, 4242, lo, 0.9)
        wkind = capi.W_DIRICHLET
    else:
        J = 6
        sizes = [N // J + 37 * j for j in range(J)]
        mix = np.random.default_rng(8).dirichlet(np.full(Kt, 0.5), J)  # per-group mixing proportions
        mine = lcd.shard_groups(sizes, world, rank)
        ctx.synth_groups([sizes[j] for j in mine], D, Kt, mu, L, 4242, mix=mix[mine], group_ids=mine)
        ctx.set_sharding(True)
        wkind = capi.W_GDIRICHLET
    if world > 1:
        ctx.set_allreduce(lcd.make_device_hook(dev))
    F, model = ctx.cluster(wkind, nthreads=4, ckind=CK)
    out = {"world": world, "F": F, "K": model.dims()[1], "rounds": model.rounds(),
           "N": [model.cluster(k)["N"] for k in range(model.dims()[1])]}
    model.close()
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
