"""Row-sharded multi-GPU plumbing: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).  The data path has a
single exchange step per EM iteration -- the all-reduce of the packed
sufficient statistics (K*(1+D+D^2) doubles + J*K counts) and of [Fz; LL_k]
(SURVEY 8(e)).  This module only moves bytes; the arithmetic stays in the
C-ABI library."""
from __future__ import annotations

import numpy as np


def shard_rows(N: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row block [lo, hi) of rank `rank` (sizes differ by at most 1)."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_groups(sizes, world: int, rank: int) -> list[int]:
    """Whole groups to ranks, greedy by size (largest first), so the per-group
    counts N_jk stay local.  Returns this rank's group indices."""
    order = sorted(range(len(sizes)), key=lambda j: (-sizes[j], j))
    load = [0] * world
    mine = []
    for j in order:
        r = min(range(world), key=lambda t: (load[t], t))
        load[r] += sizes[j]
        if r == rank:
            mine.append(j)
    return sorted(mine)


def pack_stats(Nk, xs, xxs, Njk) -> np.ndarray:
    """The all-reduced buffer layout: per cluster [N_k, s_k, S_k], then the J x K counts."""
    K = len(Nk)
    rec = np.concatenate([np.asarray(Nk).reshape(K, 1), np.asarray(xs).reshape(K, -1),
                          np.asarray(xxs).reshape(K, -1)], axis=1)
    return np.concatenate([rec.ravel(), np.asarray(Njk).ravel()])


def unpack_stats(buf, K: int, D: int, J: int):
    rec = buf[: K * (1 + D + D * D)].reshape(K, 1 + D + D * D)
    return rec[:, 0].copy(), rec[:, 1:1 + D].copy(), rec[:, 1 + D:].reshape(K, D, D).copy(), \
        buf[K * (1 + D + D * D):].reshape(J, K).copy()


def allreduce_numpy(buf: np.ndarray) -> np.ndarray:
    """Sum a host buffer across ranks (gloo path of the CPU tests)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    dist.all_reduce(t)
    return t.numpy()


class _DeviceSpan:
    """View of `n` doubles of raw device memory for torch (no copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_device_hook(device_index: int):
    """All-reduce hook for capi.Context.set_allreduce: RCCL sum, in place, on the
    device buffer the C-ABI hands over, ordered on torch's current stream (which
    must be the context's stream)."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", device_index)
    on_device = dist.get_backend() == "nccl"

    def hook(ptr: int, count: int, stream: int) -> None:
        t = torch.as_tensor(_DeviceSpan(ptr, count), device=dev)
        if on_device:
            dist.all_reduce(t)  # RCCL, in place on the C-ABI's buffer, ordered on the current stream
        else:
            # gloo (CPU collectives; used to exercise the multi-rank path where RCCL cannot run,
            # e.g. several ranks on one GPU): bounce through the host
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

    return hook
