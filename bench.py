#!/usr/bin/env python3
"""Benchmark of the libcluster E-step hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full VBEM iteration of the reference loop (cluster.cpp:198-234)
at fixed K on a synthetic full-covariance Gaussian mixture resident in HBM:
suff-stat kernel (updateSS/addobs) -> [N>1: RCCL all-reduce] -> host M-step ->
E-step kernel (vbexpectation) -> [N>1: all-reduce of Fz] -> free energy.
Nothing is skipped inside the timed region.  value = rows processed by all
ranks per second (weak scaling: every GPU holds `N` rows).

Default workload: the configuration BASELINE.json's north_star quotes the
metric on -- BGMM, N=10M rows per GPU, D=64, K=32 (configs[3] is the same
shape at 8 GPUs).  --config 2 / 3 select BASELINE.json configs[1] / configs[2].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix spec; v_mfma_f64_4x4x4 measured 73.9 (profiles/)

CONFIGS = {
    # name: (N per GPU, D, K, weight kind, seed)
    "northstar": dict(N=10_000_000, D=64, K=32, w="Dirichlet", seed=1004,
                      label="BGMM N=10M/GPU D=64 K=32 (north_star target; BASELINE configs[3] = 8 of these)"),
    "2": dict(N=1_000_000, D=16, K=8, w="Dirichlet", seed=1002, label="BASELINE configs[1]: BGMM N=1M D=16 K=8"),
    "3": dict(N=10_000_000, D=64, K=32, w="StickBreak", seed=1003,
              label="BASELINE configs[2]: VDP N=10M D=64 K=32"),
    "5": dict(N=4_000_000, D=128, K=64, w="GDirichlet", seed=1005, J=8,
              label="BASELINE configs[4] per GPU: GMC, 8 groups x 500k rows, D=128 K=64 (64 groups on 8 GPUs)"),
    "tiny": dict(N=200_000, D=16, K=4, w="Dirichlet", seed=7, label="smoke: N=200k D=16 K=4"),
    # the diagonal / exponential families at the north-star shape (HBM/VALU-bound kernels)
    "dgmm": dict(N=10_000_000, D=64, K=32, w="Dirichlet", c="NormGamma", seed=1006,
                 label="DGMM (diagonal Gaussians, NormGamma) N=10M/GPU D=64 K=32"),
    "bemm": dict(N=10_000_000, D=64, K=32, w="Dirichlet", c="ExpGamma", seed=1007,
                 label="BEMM (exponential clusters, ExpGamma) N=10M/GPU D=64 K=32"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E (/opt/skills/guides/MI355X_MICROARCH.md)


def mixture(D, K, seed, family="GaussWish"):
    """mu_k ~ N(0, 9 I), Sigma_k = B B^T / D + 0.5 I.  Diagonal families: axis-aligned components
    (ExpGamma needs x >= 0: means 20..60, unit-scale spread)."""
    rng = np.random.default_rng(seed)
    if family != "GaussWish":
        mu = rng.normal(0.0, 3.0, (K, D)) if family == "NormGamma" else rng.uniform(20.0, 60.0, (K, D))
        return mu, np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
    mu = rng.normal(0.0, 3.0, (K, D))
    L = np.empty((K, D, D))
    for k in range(K):
        B = rng.normal(size=(D, D))
        L[k] = np.linalg.cholesky(B @ B.T / D + 0.5 * np.eye(D))
    return mu, L


def alg_flops(N, D, K):
    """Algorithmic (minimal, structure-exploiting) flops per launch, SURVEY 8(d)."""
    return {"estep": N * K * (D * D + 4 * D), "suffstat": N * K * (D * D + 3 * D + 1)}


def cpu_baseline_family(ctx, cfg, sample_rows):
    """Diagonal / exponential families: the numpy oracle's own VBEM iteration (updateSS + update + vbexpectation,
    oracle/lc_oracle.py) on the first rows of the same stream, one thread."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import lc_oracle as o
    from threadpoolctl import threadpool_limits

    n = min(sample_rows, cfg["N"])
    X = [ctx.get_rows(0, 0, n)]
    q0 = [ctx.get_qz_rows(0, 0, n)]
    cf = o.NormGamma if cfg["c"] == "NormGamma" else o.ExpGamma
    with threadpool_limits(limits=1):
        o.vbem_fixed(X, q0, o.Dirichlet, 1.0, 1, False, cf)
        t0 = time.perf_counter()
        o.vbem_fixed(X, q0, o.Dirichlet, 1.0, 2, False, cf)
        dt = (time.perf_counter() - t0) / 2
    return {"value": n / dt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"first {n} rows of the same synthetic stream, one full VBEM iteration of oracle/lc_oracle.py "
                      f"(numpy, 1 thread), mean of 2"}


def cpu_baseline(ctx, model, cfg, wkind_name, sample_rows):
    """Time the C port of the reference arithmetic (oracle/lc_oracle_c.c) on the
    first `sample_rows` rows of the same workload, on this host's cores."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import lc_oracle_c as oc

    D, K = cfg["D"], cfg["K"]
    n = min(sample_rows, cfg["N"] // cfg.get("J", 1))  # rows of group 0
    X = ctx.get_rows(0, 0, n)
    cl = [model.cluster(k) for k in range(K)]
    el, _ = model.weights(0)
    post = ([c["nu"] for c in cl], [c["beta"] for c in cl], np.stack([c["m"] if "m" in c else c["mean"] for c in cl]),
            np.stack([c["iW"] for c in cl]), [c["logdW"] for c in cl], el)
    cores = oc.physical_cores()
    out = {}
    for label, nt, rows in (("all_cores", cores, n), ("one_thread", 1, max(1000, n // 32))):
        Xs = X[:rows]
        q, _ = oc.estep(Xs, *post, nthreads=nt)  # warm-up (also produces the q the suff-stat pass consumes)
        oc.suffstat(Xs, q, nt)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            q, _ = oc.estep(Xs, *post, nthreads=nt)
            oc.suffstat(Xs, q, nt)
            ts.append(time.perf_counter() - t0)
        out[label] = {"pts_per_s": rows / float(np.median(ts)), "rows": rows, "threads": nt}
    return {
        "value": out["all_cores"]["pts_per_s"], "unit": "points/s", "cores": cores, "kind": "port",
        "sample": f"first {n} rows of the same synthetic stream, E-step + suff-stats per cluster pass "
                  f"(oracle/lc_oracle_c.c, gcc -O3 -march=native -fopenmp, rows chunked over {cores} threads), "
                  f"median of 3",
        "one_thread_value": out["one_thread"]["pts_per_s"],
        "one_thread_sample_rows": out["one_thread"]["rows"],
    }


def group_mix(cfg, gids):
    """Per-group mixing proportions Dir(0.5 * 1_K), a function of the GLOBAL group id."""
    return np.stack([np.random.default_rng([cfg["seed"], int(g)]).dirichlet(np.full(cfg["K"], 0.5)) for g in gids])


def parity(capi, cfg, wkind, mu, L, device, rows=20000, iters=3, ckind=0):
    """Free-energy / qZ delta of the GPU path vs the numpy oracle on identical inputs
    (the first `rows` rows of the Philox stream of every group, same initial qZ)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import lc_oracle as o

    D, K, J = cfg["D"], cfg["K"], cfg.get("J", 1)
    nj = [rows // J] * J
    with capi.Context(device) as c2:
        if J == 1:
            c2.synth(rows, D, K, mu, L, cfg["seed"], 0, 0.9)
        else:
            c2.synth_groups(nj, D, K, mu, L, cfg["seed"], mix=group_mix(cfg, range(J)), group_ids=list(range(J)))
        X = [c2.get_rows(j, 0, nj[j]) for j in range(J)]
        q0 = c2.get_qz(nj)
        F, tr, m = c2.vbem(wkind, fixed_iters=iters, nthreads=8, ckind=ckind)
        q = c2.get_qz(nj)
        m.close()
    wf = {"Dirichlet": o.Dirichlet, "StickBreak": o.StickBreak, "GDirichlet": o.GDirichlet}[cfg["w"]]
    cf = {"GaussWish": o.GaussWish, "NormGamma": o.NormGamma, "ExpGamma": o.ExpGamma}[cfg.get("c", "GaussWish")]
    Ftr, _, qT, _, _ = o.vbem_fixed(X, q0, wf, 1.0, iters, False, cf)
    q, qT = np.vstack(q), np.vstack(qT)
    big = qT > 1e-12
    return {
        "rows": rows, "iters": iters, "F_gpu": float(tr[-1]), "F_cpu": float(Ftr[-1]),
        "rel_dF": float(abs(tr[-1] - Ftr[-1]) / abs(Ftr[-1])),
        "max_rel_dqZ": float(np.max(np.abs(q[big] - qT[big]) / qT[big])),
        "max_abs_dqZ": float(np.max(np.abs(q - qT))),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="northstar", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch

    if os.environ.get("LC_ALL_RANKS_ON_GPU0"):  # multi-rank smoke test on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("LC_DIST_BACKEND", "nccl")  # "gloo": test several ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from libcluster_amd import build, capi
    from libcluster_amd import dist as lcd

    if not capi.LIB_PATH.exists():
        build.build()
    cfg = dict(CONFIGS[args.config])
    if args.rows:
        cfg["N"] = args.rows
    N, D, K = cfg["N"], cfg["D"], cfg["K"]
    wkind = {"Dirichlet": capi.W_DIRICHLET, "StickBreak": capi.W_STICKBREAK, "GDirichlet": capi.W_GDIRICHLET}[cfg["w"]]
    J = cfg.get("J", 1)
    family = cfg.get("c", "GaussWish")
    ckind = {"GaussWish": capi.C_GAUSSWISH, "NormGamma": capi.C_NORMGAMMA, "ExpGamma": capi.C_EXPGAMMA}[family]
    mu, L = mixture(D, K, cfg["seed"], family)
    nthreads = max(1, min(32, (os.cpu_count() or 2) // max(1, world)))

    stream = torch.cuda.current_stream().cuda_stream
    ctx = capi.Context(local_rank, stream)
    if J == 1:
        ctx.synth(N, D, K, mu, L, cfg["seed"], rank * N, 0.9)  # this rank's row block of the one stream
    else:
        gids = list(range(rank * J, (rank + 1) * J))  # whole groups per rank
        ctx.synth_groups([N // J] * J, D, K, mu, L, cfg["seed"], mix=group_mix(cfg, gids), group_ids=gids)
        ctx.set_sharding(True)
    if world > 1:
        ctx.set_allreduce(lcd.make_device_hook(local_rank))

    model = None
    if args.warmup > 0:
        _, _, model = ctx.vbem(wkind, fixed_iters=args.warmup, nthreads=nthreads, ckind=ckind)
    ctx.timing_enable(True)
    ctx.timing_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F, tr, model = ctx.vbem(wkind, fixed_iters=args.steps, nthreads=nthreads, model=model, ckind=ckind)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kt = ctx.timing_get()
    ctx.timing_enable(False)

    if rank == 0:
        steps = args.steps
        fl = alg_flops(N, D, K)
        est = kt["estep_ms"] / max(1, kt["estep_calls"])
        sst = kt["suffstat_ms"] / max(1, kt["suffstat_calls"])
        dom = "estep_kernel" if est >= sst else "suffstat_kernel"
        if family != "GaussWish":
            dom = dom.replace("_kernel", "_diag_kernel")
        dom_ms = max(est, sst)
        dom_fl = fl["estep"] if est >= sst else fl["suffstat"]
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        both = (fl["estep"] + fl["suffstat"]) / ((est + sst) * 1e-3) / 1e12 if est + sst > 0 else 0.0
        traffic = None  # HBM bytes per launch of the dominant kernel, from the committed PMC passes (not live)
        tf = ROOT / "profiles" / "r01_pmc_traffic.json"
        if tf.exists() and not args.rows:
            traffic = json.loads(tf.read_text()).get(args.config, {}).get(dom)
        line = {
            "metric": "E-step data-points/sec (full VBEM iteration: suff-stats + M-step + E-step)",
            "value": world * N * steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg["label"], "rows_per_gpu": N, "D": D, "K": K, "weights": cfg["w"],
                       "seed": cfg["seed"], "groups_per_gpu": J,
                       "parallelism": (f"rows sharded x{world}" if J == 1 else f"whole groups sharded x{world}")
                       + ", all-reduce of suff-stats"},
            "free_energy": float(F),
            "kernels": {"estep_ms": est, "suffstat_ms": sst, "estep_calls": kt["estep_calls"],
                        "suffstat_calls": kt["suffstat_calls"],
                        "estep_kernel_points_per_s": N / (est * 1e-3) if est > 0 else None,
                        "both_kernels_alg_tflops": both},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "alg_flops_per_launch": dom_fl, "avg_launch_ms": dom_ms},
        }
        if family != "GaussWish":
            # separable families: 8 (D + K) algorithmic bytes per row and launch (X read + q column written / read)
            gbs = 8.0 * N * (D + K) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            line["config"]["clusters"] = family
            line["kernels"].pop("both_kernels_alg_tflops")
            line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                                "alg_bytes_per_launch": 8.0 * N * (D + K), "avg_launch_ms": dom_ms}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline(ctx, model, cfg, cfg["w"], args.cpu_sample_rows)
                                    if family == "GaussWish" else
                                    cpu_baseline_family(ctx, cfg, min(args.cpu_sample_rows, 400_000)))
        if world == 1 and not args.no_parity:
            line["parity"] = parity(capi, cfg, wkind, mu, L, local_rank, ckind=ckind)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
