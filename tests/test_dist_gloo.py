"""The N>1 path on CPU: two gloo ranks, rows sharded, one all-reduce of the
packed sufficient statistics and one of [Fz] per EM iteration, host M-step
through the C-ABI on every rank.  The per-shard E-step / suff-stat arithmetic
is supplied by the oracle here (there is no GPU in this suite, and the product
has no CPU data path); what is under test is the sharding, packing, reduction
and replicated M-step plumbing of libcluster_amd.dist."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, D, K, iters, out):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    import lc_oracle as o
    from libcluster_amd import capi
    from libcluster_amd import dist as lcd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)  # same stream on every rank; each keeps its shard
        X = rng.normal(size=(N, D)) * 1.5 + rng.integers(0, K, (N, 1)) * 2.0
        q = rng.dirichlet(np.ones(K), N)
        lo, hi = lcd.shard_rows(N, world, rank)
        Xs, qs = X[lo:hi], q[lo:hi]
        Ftrace = []
        for _ in range(iters):
            # local pass (oracle stands in for the suff-stat kernel)
            Nk, xs, xxs = o.suffstats(Xs, qs)
            buf = lcd.allreduce_numpy(lcd.pack_stats(Nk, xs, xxs, Nk[None, :]))
            Nk, xs, xxs, Njk = lcd.unpack_stats(buf, K, D, 1)
            # replicated host M-step through the C-ABI
            elog, Fw = capi.weights_update(capi.W_DIRICHLET, Njk[0])
            post = [capi.gw_mstep(1.0, Nk[k], xs[k], xxs[k]) for k in range(K)]
            # local E-step (oracle stands in for the E-step kernel), then reduce Fz
            cl = []
            for k in range(K):
                g = o.GaussWish(1.0, D)
                g.nu, g.beta, g.m, g.iW, g.logdW, g.N = (post[k]["nu"], post[k]["beta"], post[k]["m"],
                                                         post[k]["iW"], post[k]["logdW"], Nk[k])
                cl.append(g)
            w = o.Dirichlet()
            w.E_logpi = elog
            w.Nk = Njk[0]
            qs, fz = o.vbexpectation(Xs, w, cl)
            Fz = float(lcd.allreduce_numpy(np.array([fz]))[0])
            Ftrace.append(Fw + sum(p["fenergy"] for p in post) + Fz)
        if rank == 0:
            ref, _, qT, _, _ = o.vbem_fixed([X], [q], o.Dirichlet, 1.0, iters)
            np.save(out, np.array([Ftrace, ref]))
        # every rank holds its rows of the final qZ
        qall = [None] * world
        dist.all_gather_object(qall, qs)
        if rank == 0:
            _, _, qT, _, _ = o.vbem_fixed([X], [q], o.Dirichlet, 1.0, iters)
            np.testing.assert_allclose(np.vstack(qall), qT[0], rtol=1e-8, atol=1e-12)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])  # (8: the world the driver's scaling run ends at; unequal shards of 62 / 63 rows)
def test_two_rank_gloo_em_matches_single_process(tmp_path, lib, world):
    import torch.multiprocessing as mp

    out = str(tmp_path / "f.npy")
    mp.spawn(_worker, args=(world, _free_port(), 501, 5, 3, 3, out), nprocs=world, join=True)
    F = np.load(out)
    np.testing.assert_allclose(F[0], F[1], rtol=1e-11)


def test_sharding_helpers():
    from libcluster_amd import dist as lcd

    for N, W in ((10, 3), (7, 8), (80_000_000, 8), (0, 2)):
        spans = [lcd.shard_rows(N, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    sizes = [500_000] * 64
    got = sorted(j for r in range(8) for j in lcd.shard_groups(sizes, 8, r))
    assert got == list(range(64))
    assert all(len(lcd.shard_groups(sizes, 8, r)) == 8 for r in range(8))
    sizes = [5, 1, 9, 3, 3, 7]
    loads = [sum(sizes[j] for j in lcd.shard_groups(sizes, 2, r)) for r in range(2)]
    assert sum(loads) == sum(sizes) and abs(loads[0] - loads[1]) <= 2
    rng = np.random.default_rng(0)
    Nk, xs, xxs, Njk = rng.random(4), rng.random((4, 3)), rng.random((4, 3, 3)), rng.random((2, 4))
    a, b, c, d = lcd.unpack_stats(lcd.pack_stats(Nk, xs, xxs, Njk), 4, 3, 2)
    assert np.array_equal(a, Nk) and np.array_equal(b, xs) and np.array_equal(c, xxs) and np.array_equal(d, Njk)


@pytest.mark.gpu
def test_allreduce_hook_on_device_buffers():
    """world_size-1 RCCL group on the GPU box: the C-ABI's all-reduce hook hands
    raw device pointers to torch.distributed; results must equal the hook-less run."""
    import torch
    import torch.distributed as dist

    from libcluster_amd import capi
    from libcluster_amd import dist as lcd

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(4)
        X = rng.normal(size=(5000, 24)) + rng.integers(0, 4, (5000, 1))  # (D > 16: the two-kernel iteration)
        q = rng.dirichlet(np.ones(4), 5000)
        res = []
        for hook in (False, True):
            with capi.Context(0, torch.cuda.current_stream().cuda_stream) as ctx:
                ctx.set_data(X)
                ctx.set_qz(q)
                calls = []
                if hook:
                    inner = lcd.make_device_hook(0)

                    def h(ptr, count, stream):
                        calls.append(count)
                        inner(ptr, count, stream)

                    ctx.set_allreduce(h)
                F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=3)
                res.append((tr, ctx.get_qz([5000])[0]))
                m.close()
                if hook:
                    K, DP = 4, 32
                    assert calls.count(K * (1 + DP * DP + DP) + K) == 3  # packed stats (D = 24 -> DP = 32) + counts
                    assert calls.count(1 + K) == 3                        # [Fz; LLk]
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
    finally:
        dist.destroy_process_group()
