#!/usr/bin/env python3
"""Benchmark of the libcluster E-step hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 1
    python bench.py --gpus 8 ...          # starts the 8 ranks itself (torch.distributed.run, one process per GPU)
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
shape at 8 GPUs).  --config 2 / 3 / 5 select BASELINE.json configs[1] / [2] / [4 per GPU].

Multi-GPU: the all-reduce of the packed statistics is the library's own RCCL collective
(lc_ctx_comm_init_rccl: ncclAllReduce(ncclDouble, ncclSum) on the context's stream); torch.distributed is only the
launcher's rendezvous (broadcast of the RCCL unique id, the barriers and the max-over-ranks of the timing).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # the CPU baseline (single-rank runs only) pins its OpenMP threads: one per physical core, neighbours adjacent
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix spec; v_mfma_f64_4x4x4 measured 73.9 (profiles/)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E (/opt/skills/guides/MI355X_MICROARCH.md)

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
    # full covariances wider than 128 columns: estep_wide_kernel + panel launches of suffstat_kernel
    "wide256": dict(N=1_000_000, D=256, K=16, w="Dirichlet", seed=1008,
                    label="BGMM, wide observations: N=1M D=256 K=16 (chunk-streamed whitener, panel statistics)"),
    # ragged cluster counts (model selection walks K = 1, 2, 3, ...: most of its rounds are NOT multiples of 32) and an
    # in-between width: the statistics kernel selection of DESIGN 1 off its best shapes
    "k20": dict(N=5_000_000, D=64, K=20, w="Dirichlet", seed=1020, label="BGMM N=5M D=64 K=20 (ragged cluster count)"),
    "k40": dict(N=5_000_000, D=64, K=40, w="Dirichlet", seed=1040, label="BGMM N=5M D=64 K=40 (ragged cluster count)"),
    "d96": dict(N=4_000_000, D=96, K=32, w="Dirichlet", seed=1096, label="BGMM N=4M D=96 K=32 (in-between width)"),
    # small and medium Gauss-Wishart shapes (round 6): the widths and cluster counts cluster() walks on every call; D = 23 is
    # the width of the reference's own data set (test/scott25.dat) -- algorithmic flops are counted on D, not on the padding
    "d32": dict(N=6_000_000, D=23, K=16, w="Dirichlet", seed=1123, label="BGMM N=6M D=23 K=16 (scott25.dat's width)"),
    "d48": dict(N=5_000_000, D=48, K=12, w="Dirichlet", seed=1148, label="BGMM N=5M D=48 K=12"),
    "k8": dict(N=4_000_000, D=64, K=8, w="Dirichlet", seed=1108, label="BGMM N=4M D=64 K=8 (few clusters)"),
}
# short runs reported under "other_configs" of the default line
OTHER_CONFIGS = ["2", "3", "5", "dgmm", "bemm", "wide256", "k20", "k40", "d96", "d32", "d48", "k8"]


def mixture(D, K, seed, family="GaussWish", overlap=False):
    """mu_k ~ N(0, 9 I), Sigma_k = B B^T / D + 0.5 I.  Diagonal families: axis-aligned components
    (ExpGamma needs x >= 0: means 20..60, unit-scale spread).
    overlap=True (parity leg only): mu_k ~ N(0, 0.09 I), Sigma_k = 0.2 B B^T / D + I -- components on top of each
    other, so that most responsibilities are soft (the 8(d) mixture is ~770 Mahalanobis^2 apart at D = 64: every
    responsibility above the comparison mask is exactly 1 and a qZ comparison cannot fail)."""
    rng = np.random.default_rng(seed)
    if family != "GaussWish":
        mu = rng.normal(0.0, 3.0, (K, D)) if family == "NormGamma" else rng.uniform(20.0, 60.0, (K, D))
        return mu, np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
    mu = rng.normal(0.0, 0.3 if overlap else 3.0, (K, D))
    L = np.empty((K, D, D))
    for k in range(K):
        B = rng.normal(size=(D, D))
        L[k] = np.linalg.cholesky((0.2 * (B @ B.T) / D + np.eye(D)) if overlap else (B @ B.T / D + 0.5 * np.eye(D)))
    return mu, L


def summary_of(line):
    """{config: [ms_per_step, frac, estep_frac, suffstat_frac, traffic / algorithmic bytes]} for the headline and every
    entry of other_configs, plus model selection and the drop-in call in seconds -- compact and LAST on the line, so that
    every configuration survives a record that keeps only the line's tail (the driver kept the last 8 KB)."""
    def r4(v):
        return None if v is None else float(f"{v:.4g}")

    def row(ms, roof):
        tr = roof.get("traffic")
        tb = None
        if isinstance(tr, dict):
            tb = tr.get("bytes_per_launch", tr.get("total_bytes"))
            if tb is None and "read_bytes" in tr:
                tb = tr["read_bytes"] + tr.get("write_bytes", 0.0)
        elif isinstance(tr, (int, float)):
            tb = float(tr)
        ratio = tb / roof["alg_bytes_per_launch"] if tb and roof.get("alg_bytes_per_launch") else None
        if roof.get("mfma_frac") is not None:  # separable families: [ms, frac, mfma_frac, hbm_frac, traffic ratio]
            return [r4(ms), r4(roof["frac"]), r4(roof["mfma_frac"]), r4(roof["hbm_frac"]), r4(ratio)]
        if str(roof.get("kernel", "")).startswith("fused"):  # one launch does both passes: `frac` is the figure
            return [r4(ms), r4(roof["frac"]), None, None, r4(ratio)]
        return [r4(ms), r4(roof["frac"]), r4(roof.get("estep_frac")), r4(roof.get("suffstat_frac")), r4(ratio)]

    out = {"_columns": "ms_per_step, roofline.frac, estep_frac (separable families: mfma_frac), suffstat_frac (hbm_frac), "
                       "PMC traffic / algorithmic bytes of the dominant kernel",
           "headline": row(line["ms_per_step"], line["roofline"])}
    for o in line.get("other_configs", []):
        out[o["config"]] = row(o["ms_per_step"], o["roofline"])
    ms = line.get("model_selection")
    if isinstance(ms, dict) and "seconds" in ms:
        out["model_selection_s"] = r4(ms["seconds"])
    dc = line.get("dropin_call")
    if isinstance(dc, dict) and "seconds" in dc:
        out["dropin_call_s"] = r4(dc["seconds"])
    return out


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


def cpu_baseline(ctx, model, cfg, sample_rows, reps=5):
    """Time the C port of the reference arithmetic (oracle/lc_oracle_c.c) on the first `sample_rows` rows of the same
    workload, on this host's cores: mode B of SURVEY 8(d) (rows chunked over all physical cores, threads pinned) and
    mode A (one thread: what learnBGMM / learnVDP really use, cluster.cpp:207-223 + the `critical` at :77-78), both on
    the SAME rows.  Dense flops per point K(4 D^2 + 5 D + 1): two triangular solves + the full D x D product."""
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
    w_dense = K * (4 * D * D + 5 * D + 1)

    def one(nt, rows):
        t0 = time.perf_counter()
        q, _ = oc.estep(X[:rows], *post, nthreads=nt)
        oc.suffstat(X[:rows], q, nt)
        return time.perf_counter() - t0

    one(cores, n)  # warm-up (page faults, thread pool)
    ts = [one(cores, n) for _ in range(reps)]
    all_pts = n / float(np.median(ts))
    one(1, min(n, 20_000))  # 1-thread warm-up on a short prefix, then ONE timed pass over the same n rows
    one_pts = n / one(1, n)
    peak, peak_desc = oc.host_fp64_peak_gflops(cores)
    gf = all_pts * w_dense / 1e9
    return {
        "value": all_pts, "unit": "points/s", "cores": cores, "kind": "port", "per_core_value": all_pts / cores,
        "cores_visible_note": (f"{cores} physical cores are visible to this process (os.sched_getaffinity: "
                               f"{len(os.sched_getaffinity(0))} logical CPUs); the gpurun / driver leases expose two, "
                               "so `value` is a two-thread figure there -- `per_core_value` is the comparable number"),
        "sample": f"first {n} rows of the same synthetic stream, E-step + suff-stats (oracle/lc_oracle_c.c, gcc -O3 "
                  f"-march=native -fopenmp, rows chunked over {cores} threads pinned with OMP_PROC_BIND=close "
                  f"OMP_PLACES=cores, X read once per 128-row tile for all K), median of {reps} after 1 warm-up",
        "dense_gflops": gf, "dense_flops_per_point": w_dense,
        "host_fp64_peak_gflops": peak, "host_fp64_peak_is": peak_desc,
        "frac_of_host_peak": (gf / peak) if peak else None,
        "one_thread_value": one_pts, "one_thread_sample_rows": n, "one_thread_dense_gflops": one_pts * w_dense / 1e9,
        "scaling_over_one_thread": all_pts / one_pts,
        "scaling_note": (f"{all_pts / one_pts:.1f}x on {cores} threads.  The one-thread run has the core's boost clock and "
                         "a whole L3 to itself; the all-core run is at the all-core clock (the nominal peak above assumes "
                         "the maximum clock on every core), shares L3 and the memory channels, and both modes stream the K "
                         f"Cholesky factors ({K * D * D * 8 // 1024} KB) from L2 once per 32-row tile in the E-step, which "
                         "bounds them at a fraction of the FMA peak; the statistics pass reads X once per 128-row tile."),
    }


def group_mix(cfg, gids):
    """Per-group mixing proportions Dir(0.5 * 1_K), a function of the GLOBAL group id."""
    return np.stack([np.random.default_rng([cfg["seed"], int(g)]).dirichlet(np.full(cfg["K"], 0.5)) for g in gids])


def _qz_delta(q, qT):
    big = qT > 1e-12
    soft = (qT > 1e-6) & (qT < 1.0 - 1e-6)
    return {"max_rel_dqZ": float(np.max(np.abs(q[big] - qT[big]) / qT[big])),
            "max_abs_dqZ": float(np.max(np.abs(q - qT))),
            "soft_entry_fraction": float(soft.mean()),
            "max_rel_dqZ_soft": float(np.max(np.abs(q[soft] - qT[soft]) / qT[soft])) if soft.any() else None}


def parity_numpy(capi, cfg, wkind, mu, L, device, rows=20000, iters=3, ckind=0):
    """Grouped workloads and the separable families: F / qZ delta of the GPU path vs the numpy oracle on identical
    inputs (the first `rows` rows of the Philox stream of every group, same initial qZ)."""
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
    out = {"checker": "oracle/lc_oracle.py (numpy)", "rows": rows, "iters": iters, "F_gpu": float(tr[-1]),
           "F_cpu": float(Ftr[-1]), "rel_dF": float(abs(tr[-1] - Ftr[-1]) / abs(Ftr[-1]))}
    out.update(_qz_delta(np.vstack(q), np.vstack(qT)))
    return out


def parity_c(capi, cfg, wkind, device, rows, iters, overlap, nthreads):
    """GPU and CPU both run the first N_par = 1e6 rows of a Philox stream from the same initial qZ for the
    same T iterations; F and qZ are compared there.  CPU side: oracle/lc_oracle_c.c (data passes, all cores) + the
    numpy oracle's M-step and free energy (lc_oracle_c.vbem_fixed; equal to the pure numpy oracle,
    tests/test_oracle_c.py).  overlap=False: the workload's own mixture (hard responsibilities -- F is the meaningful
    figure); overlap=True: components on top of each other (see mixture()), where most entries of qZ are soft."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import lc_oracle as o
    import lc_oracle_c as oc

    D, K = cfg["D"], cfg["K"]
    seed = cfg["seed"] + (7919 if overlap else 0)
    mu, L = mixture(D, K, seed, "GaussWish", overlap)
    with capi.Context(device) as c2:
        c2.synth(rows, D, K, mu, L, seed, 0, 0.9)
        X = c2.get_rows(0, 0, rows)
        q0 = c2.get_qz_rows(0, 0, rows)
        F, tr, m = c2.vbem(wkind, fixed_iters=iters, nthreads=8)
        q = c2.get_qz_rows(0, 0, rows)
        m.close()
    wf = {"Dirichlet": o.Dirichlet, "StickBreak": o.StickBreak, "GDirichlet": o.GDirichlet}[cfg["w"]]
    Ftr, qT = oc.vbem_fixed(X, q0, wf, 1.0, iters, nthreads)
    out = {"checker": "oracle/lc_oracle_c.c + numpy M-step (lc_oracle_c.vbem_fixed)", "rows": rows, "iters": iters,
           "mixture": ("overlapping: mu ~ N(0, 0.09 I), Sigma = 0.2 B B^T / D + I" if overlap
                       else "the workload's (SURVEY 8(d)): mu ~ N(0, 9 I), Sigma = B B^T / D + 0.5 I"),
           "F_gpu": float(tr[-1]), "F_cpu": float(Ftr[-1]),
           "rel_dF": float(abs(tr[-1] - Ftr[-1]) / abs(Ftr[-1])),
           "max_rel_dF_trace": float(np.max(np.abs(np.array(tr) - np.array(Ftr)) / np.abs(Ftr)))}
    out.update(_qz_delta(q, qT))
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (one process per GPU) BEFORE this
    process touches the GPU, wait, and pass rank 0's line through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.pop("OMP_PROC_BIND", None)  # (N ranks must not all pin their threads to the same cores)
    env.pop("OMP_PLACES", None)
    return subprocess.run(cmd, env=env).returncode


def setup_comm(ctx, capi, lcd, dist, torch, rank, world, local_rank, want):
    """The statistics all-reduce of a multi-rank run.  "native": the library's own collective -- RCCL (one rank per GPU)
    or, with the gloo test backend (several ranks on one GPU), its host-staged transport; checked with a known sum on
    every rank before it is trusted, else (or with --comm torch) the Python hook over torch.distributed."""
    kind, note = None, None
    if want == "native":
        ok = 1.0
        try:
            if dist.get_backend() == "nccl":
                idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                ctx.comm_init_rccl(bytes(idt.cpu().numpy().tobytes()), rank, world)
            else:
                ctx.comm_init_host(f"bench_{os.environ.get('MASTER_PORT', '0')}", rank, world)
            got = ctx.allreduce([rank + 1.0, 1.0, -0.25 * (rank + 1)])
            tot = 0.5 * world * (world + 1)
            if not np.array_equal(got, [tot, float(world), -0.25 * tot]):
                raise RuntimeError(f"all-reduce self-check returned {got.tolist()}")
            kind = ctx.comm_info()["kind"]
        except Exception as e:  # noqa: BLE001
            ok, note = 0.0, f"native collective unavailable on rank {rank}: {e}"
        flag = torch.tensor([ok], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:  # every rank falls back together
            try:
                ctx.comm_free()
            except Exception:  # noqa: BLE001
                pass
            kind = None
    if kind is None:
        ctx.set_allreduce(lcd.make_device_hook(local_rank))
        kind = "hook (torch.distributed " + dist.get_backend() + ")"
    return kind, note


def model_selection(capi, device, stream, N=10_000_000, D=64, Kt=32):
    """The whole learnVDP loop (cluster.cpp:564-629: VBEM, prune, greedy split search) on device-resident synthetic
    observations -- the caller of the hot path; tools/learn_bench.py is the same measurement."""
    import numpy as np

    rng = np.random.default_rng(5)
    mu = rng.normal(0, 4.0, (Kt, D))
    L = np.stack([np.linalg.cholesky((lambda B: B @ B.T / D + 0.5 * np.eye(D))(rng.normal(size=(D, D)))) for _ in range(Kt)])
    with capi.Context(device, stream) as ctx:
        # one untimed run at full size first (as the W warm-up steps of the main measurement): it loads the kernels and
        # maps the loop's buffers (distance slab, moves, responsibilities: ~8 GB whose first touch costs 0.5-0.8 s on a
        # fresh process and depends on what the block cache holds); the timed run is the second one
        ctx.synth(N, D, Kt, mu, L, 99, 0, 0.9)
        ctx.cluster(capi.W_STICKBREAK, nthreads=16)[1].close()
        ctx.synth(N, D, Kt, mu, L, 99, 0, 0.9)
        ctx.synchronize()
        t0 = time.perf_counter()
        F, model = ctx.cluster(capi.W_STICKBREAK, nthreads=16)
        dt = time.perf_counter() - t0
        rounds = model.rounds()
        K = model.dims()[1]
        model.close()
    return {"workload": f"learnVDP N={N} D={D}, {Kt} true clusters, from one cluster up (device-resident data)",
            "seconds": dt, "warmup": "one untimed run of the same loop", "K_found": K, "rounds": len(rounds),
            "main_vbem_iterations": sum(len(t) for _, t in rounds),
            "free_energy": F}


def dropin_call(capi, device, N=10_000_000, D=64, Kt=32):
    """The call a libcluster user makes (cluster.cpp:636-664 `learnVDP(X, qZ, weights, clusters)`, python/libclusterpy.cpp:
    135-158): ONE learnVDP on a HOST matrix -- upload, the whole model selection on the device, the responsibilities and
    the posterior back -- timed end to end through libcluster_amd.learnVDP (ctypes over the C ABI, what the C++ facade of
    include/libcluster.h calls too).  Plus the two PCIe legs on their own: lc_ctx_set_data of the same pageable array and
    lc_ctx_get_qz_all of an N x K matrix, with the GB/s they reach.  Host memory: X 5.1 GB + qZ 2.6 GB."""
    import libcluster_amd as lc

    rng = np.random.default_rng(11)
    mu = rng.normal(0, 4.0, (Kt, D))
    sd = rng.uniform(0.7, 1.3, (Kt, 1))
    X = np.empty((N, D))
    piece = 1_000_000
    for r0 in range(0, N, piece):  # (in pieces: no N x D temporaries next to X)
        z = rng.integers(0, Kt, min(piece, N - r0))
        X[r0:r0 + z.size] = mu[z] + rng.standard_normal((z.size, D)) * sd[z]
    nthreads = max(1, min(32, len(os.sched_getaffinity(0))))
    lc.learnVDP(X[:200_000], nthreads=nthreads, device=device)  # (loads the kernels; the timed call is the full one)
    t0 = time.perf_counter()
    F, qZ, w, means, covs, info = lc.learnVDP(X, nthreads=nthreads, device=device, return_info=True)
    dt = time.perf_counter() - t0
    K = info["K"]
    out = {"workload": f"learnVDP(X) on a host numpy matrix N={N} D={D} ({Kt} true clusters): upload + model selection + "
                       "qZ / weights / means / covariances back (libcluster_amd.learnVDP over the C ABI)",
           "seconds": dt, "K_found": K, "rounds": len(info["rounds"]), "free_energy": float(F),
           "qZ_shape": list(qZ.shape), "row_sums_ok": bool(np.allclose(qZ[:: max(1, N // 1000)].sum(1), 1.0, atol=1e-9))}
    del qZ, w, means, covs, info
    with capi.Context(device) as ctx:
        ctx.set_data(X[:1000])  # (page-locked staging buffers exist from here on)
        t0 = time.perf_counter()
        ctx.set_data(X)
        ctx.synchronize()
        up = time.perf_counter() - t0
        ctx.fill_qz(K, 1.0 / K)
        ctx.synchronize()
        t0 = time.perf_counter()
        q = ctx.get_qz([N])[0]
        down = time.perf_counter() - t0
        out.update({"h2d_seconds": up, "h2d_GBps": X.nbytes / up / 1e9, "d2h_seconds": down,
                    "d2h_GBps": q.nbytes / down / 1e9, "h2d_bytes": int(X.nbytes), "d2h_bytes": int(q.nbytes),
                    "pcie_note": "pageable numpy arrays, row-major; packed / unpacked through two page-locked 32 MB buffers "
                                 "(lc_ctx_set_data / lc_ctx_get_qz_all), d2h includes the device-side transpose and the "
                                 "allocation of the result"})
    return out


def measure(capi, cfg, steps, warmup, rank, world, local_rank, stream, nthreads, comm=None, dist=None, torch=None):
    """Synthesise the workload in HBM, run `warmup` untimed and `steps` timed VBEM iterations; returns everything the
    JSON line needs plus the live context / model (for the CPU baseline)."""
    N, D, K = cfg["N"], cfg["D"], cfg["K"]
    wkind = {"Dirichlet": capi.W_DIRICHLET, "StickBreak": capi.W_STICKBREAK, "GDirichlet": capi.W_GDIRICHLET}[cfg["w"]]
    J = cfg.get("J", 1)
    family = cfg.get("c", "GaussWish")
    ckind = {"GaussWish": capi.C_GAUSSWISH, "NormGamma": capi.C_NORMGAMMA, "ExpGamma": capi.C_EXPGAMMA}[family]
    mu, L = mixture(D, K, cfg["seed"], family)
    ctx = capi.Context(local_rank, stream)
    if J == 1 and D > 128:
        # (the on-device generator stops at D = 128) the same mixture drawn on the host and uploaded once, outside the
        # timed region; initial responsibilities = smoothed labels, as lc_ctx_synth leaves them
        rng = np.random.default_rng([cfg["seed"], rank])
        z = rng.integers(0, K, N)
        X = np.empty((N, D))
        for k in range(K):
            idx = np.flatnonzero(z == k)
            X[idx] = mu[k] + rng.standard_normal((idx.size, D)) @ L[k].T
        q0 = np.full((N, K), 0.1 / (K - 1))
        q0[np.arange(N), z] = 0.9
        ctx.set_data(X)
        ctx.set_qz(q0)
        del X, q0
    elif J == 1:
        ctx.synth(N, D, K, mu, L, cfg["seed"], rank * N, 0.9)  # this rank's row block of the one stream
    else:
        gids = list(range(rank * J, (rank + 1) * J))  # whole groups per rank
        ctx.synth_groups([N // J] * J, D, K, mu, L, cfg["seed"], mix=group_mix(cfg, gids), group_ids=gids)
        ctx.set_sharding(True)
    comm_kind = comm_note = None
    if world > 1 or (dist is not None and comm is not None):
        comm_kind, comm_note = comm(ctx)

    model = None
    if warmup > 0:
        _, _, model = ctx.vbem(wkind, fixed_iters=warmup, nthreads=nthreads, ckind=ckind)
    ctx.timing_enable(True)
    ctx.timing_reset()
    if dist is not None:
        dist.barrier()
    if torch is not None:
        torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    F, tr, model = ctx.vbem(wkind, fixed_iters=steps, nthreads=nthreads, model=model, ckind=ckind)
    if torch is not None:
        torch.cuda.synchronize()
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    dt = own_dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kt = ctx.timing_get()
    ka = ctx.timing_get_all()
    ctx.timing_enable(False)
    # what THIS rank spent where, per step (ms): its two kernels, the exchange step (the sum + the wait for the slowest
    # rank), the host M-step, and the host's remaining wall time in the two data-pass phases (launches, copies,
    # synchronisation) -- enough to tell a slow collective from a slow M-step from a straggler on a first 8-GPU run
    it = max(1, ka["host_iters"])
    dev_ms = (ka["estep_ms"] + ka["suffstat_ms"] + ka["fused_ms"] + ka["allreduce_ms"]) / it
    rank_report = {
        "rank": rank, "device": local_rank, "step_ms": own_dt / steps * 1e3,
        "estep_ms": ka["estep_ms"] / max(1, ka["estep_calls"]), "suffstat_ms": ka["suffstat_ms"] / max(1, ka["suffstat_calls"]),
        **({"fused_ms": ka["fused_ms"] / ka["fused_calls"]} if ka["fused_calls"] else {}),
        "allreduce_ms": ka["allreduce_ms"] / it, "allreduce_calls_per_step": ka["allreduce_calls"] / it,
        # host phases of an iteration (vbem's own clocks): cluster M-step + E-step constants; the free-energy tail; the
        # statistics phase's host share (weights update, unpacking) is inside wait_ms with the launches and the waits
        "mstep_ms": ka["host_mstep_ms"] / it, "fenergy_ms": ka["host_fenergy_ms"] / it,
        "wait_ms": max(0.0, (ka["host_stats_ms"] + ka["host_estep_ms"]) / it - dev_ms),
        "mstep_threads": nthreads, "cpus": len(os.sched_getaffinity(0)),
    }
    per_rank = [rank_report]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_report)

    fl = alg_flops(N, D, K)
    est = kt["estep_ms"] / max(1, kt["estep_calls"])
    sst = kt["suffstat_ms"] / max(1, kt["suffstat_calls"])
    ssname = "suffstat_kernel"
    if family == "GaussWish":
        fn = capi.lib().lc_statistics_kernel_name  # per-cluster form or the feature GEMM: what rocprofv3 will list
        fn.restype = __import__("ctypes").c_char_p
        fn.argtypes = [__import__("ctypes").c_int, __import__("ctypes").c_int]
        ssname = fn(D, K).decode()
    dom = ("estep_wide_kernel" if D > 128 else "estep_kernel") if est >= sst else ssname
    if family != "GaussWish":  # the names rocprofv3 lists: the matrix-pipe E-step where the context took it (every launch here)
        dom = ("estep_diag_mfma_kernel" if ka.get("estep_diag_mfma_calls", 0) > 0 else "estep_diag_kernel") if est >= sst \
            else "suffstat_diag_kernel"
    dom_ms = max(est, sst)
    dom_fl = fl["estep"] if est >= sst else fl["suffstat"]
    fused = kt.get("fused_calls", 0)
    if fused:  # small observations: one launch does the E-step and the next iteration's statistics
        fus = kt["fused_ms"] / fused
        dom, dom_ms, dom_fl = "fused_small_kernel", fus, fl["estep"] + fl["suffstat"]
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    both = (fl["estep"] + fl["suffstat"]) / ((est + sst) * 1e-3) / 1e12 if est + sst > 0 else 0.0
    res = {
        "value": world * N * steps / dt, "ms_per_step": dt / steps * 1e3, "free_energy": float(F),
        "kernels": {"estep_ms": est, "suffstat_ms": sst, "estep_calls": kt["estep_calls"],
                    "suffstat_calls": kt["suffstat_calls"],
                    "estep_kernel_points_per_s": N / (est * 1e-3) if est > 0 else None,
                    "both_kernels_alg_tflops": both,
                    **({"fused_ms": kt["fused_ms"] / fused, "fused_calls": fused} if fused else {})},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None,
                     "alg_flops_per_launch": dom_fl, "avg_launch_ms": dom_ms,
                     "estep_frac": fl["estep"] / (est * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if est > 0 else None,
                     "suffstat_frac": fl["suffstat"] / (sst * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if sst > 0 else None,
                     # the other roof of the same launch (report both): X read once + one q column per
                     # cluster written (E-step, fused pass) or read (statistics) = 8 N (D + K) algorithmic bytes
                     "alg_bytes_per_launch": 8.0 * N * (D + K),
                     "hbm_frac": 8.0 * N * (D + K) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else None},
        "config": {"workload": cfg["label"], "rows_per_gpu": N, "D": D, "K": K, "weights": cfg["w"],
                   "seed": cfg["seed"], "groups_per_gpu": J,
                   "parallelism": (f"rows sharded x{world}" if J == 1 else f"whole groups sharded x{world}")
                   + ", all-reduce of suff-stats"},
    }
    # where the host's wall time of a step goes (vbem's own clocks, ms per step): the statistics phase and the E-step phase
    # include the wait for their kernels; mstep = cluster M-step + the E-step's constants, fenergy = the free-energy tail
    res["host_phases_ms"] = {"stats_wall": ka["host_stats_ms"] / it, "mstep": ka["host_mstep_ms"] / it,
                             "estep_wall": ka["host_estep_ms"] / it, "fenergy": ka["host_fenergy_ms"] / it,
                             "kernels": dev_ms, "mstep_threads": nthreads, "cpus": len(os.sched_getaffinity(0))}
    if family != "GaussWish":
        # separable families: 8 (D + K) algorithmic bytes per row and launch (X read + q column written / read)
        # ... and BOTH passes are products on the matrix pipe: E-step [N x fD] . [fD x K], statistics
        # Q^T [X | X^2], f = 2 feature blocks for the diagonal Gaussians (x', x'^2), 1 for the exponential family: 2 N f D K
        # flop per launch.  The roof is whichever demand is the larger at its peak; both fractions are printed.
        nbytes = 8.0 * N * (D + K)
        mflop = 2.0 * N * (2 if family == "NormGamma" else 1) * D * K
        gbs = nbytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        tfl = mflop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        hbm_floor, mfma_floor = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3, mflop / (FP64_PEAK_TFLOPS * 1e12) * 1e3
        res["config"]["clusters"] = family
        res["kernels"].pop("both_kernels_alg_tflops")
        both = {"hbm_frac": gbs / HBM_PEAK_GBS, "mfma_frac": tfl / FP64_PEAK_TFLOPS, "alg_bytes_per_launch": nbytes,
                "alg_flops_per_launch": mflop, "avg_launch_ms": dom_ms, "hbm_floor_ms": hbm_floor, "mfma_floor_ms": mfma_floor,
                # the launch cannot be shorter than the larger of the two demands (the pipe also carries the pass's VALU
                # work -- subtractions, squares, exponentials: ~0.3 ms at the north-star shape, DESIGN 4.6 -- not counted here)
                "combined_floor_ms": max(hbm_floor, mfma_floor),
                "floor_frac": max(hbm_floor, mfma_floor) / dom_ms if dom_ms > 0 else None,
                "estep_ms": est, "suffstat_ms": sst}
        if mfma_floor > hbm_floor:
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tfl / FP64_PEAK_TFLOPS, "traffic": None, **both}
        else:
            res["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs / HBM_PEAK_GBS, "traffic": None, **both}
    if model is not None and rank == 0 and family == "GaussWish":
        # what the M-step made of the REDUCED statistics (N_k, the posterior means' and scatter matrices' sums, log|W_k|):
        # a sharded run and the one-rank run over the same rows must agree on these (tests/test_gpu_comm.py, 8 ranks)
        cl = [model.cluster(k) for k in range(K)]
        res["check"] = {"Nk": [c["N"] for c in cl], "mean_sum": [float(np.sum(c["mean"])) for c in cl],
                        "iW_trace": [float(np.trace(c["iW"])) for c in cl], "logdW": [c["logdW"] for c in cl]}
    if comm_kind:
        res["config"]["collective"] = comm_kind
        if comm_note:
            res["config"]["collective_note"] = comm_note
    if world > 1 or comm_kind:
        res["config"]["per_rank"] = per_rank
    return res, ctx, model, (wkind, ckind, mu, L, family)


def main_inproc(args):
    """`bench.py --gpus N --inproc`: the N ranks are host threads of THIS process, one context (own stream, own M-step
    pool) per GPU and the library's own collective between them -- what LIBCLUSTER_GPUS=N does inside learnBGMM /
    learnVDP / learnGMC (lc_capi.cpp learn_sharded).  RCCL (ncclCommInitRank from every thread) when the box has N
    GPUs, the host-staged transport with all shards on GPU 0 otherwise (RCCL refuses two ranks on one device).  Same
    timing contract: W untimed + K timed steps between thread barriers, the MAX over ranks, one JSON line."""
    import threading

    import torch  # noqa: F401  (one HIP runtime per process: capi binds to the copy torch loads)

    from libcluster_amd import build, capi

    if not capi.LIB_PATH.exists():
        build.build()
    world = args.gpus
    cfg = dict(CONFIGS[args.config])
    if args.rows:
        cfg["N"] = args.rows
    if args.groups:
        cfg["J"] = args.groups
    ndev = torch.cuda.device_count()
    rccl = ndev >= world and world > 1 and args.comm == "native" and not os.environ.get("LC_ALL_RANKS_ON_GPU0")
    uid = capi.comm_unique_id() if rccl else None
    name = f"bench_inproc_{os.getpid()}"
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // world)
    nthreads = int(os.environ.get("LC_BENCH_THREADS", max(1, min(32, per))))
    bar = threading.Barrier(world)
    out, errs = [None] * world, [None] * world

    class ThreadDist:  # the two things measure() asks of torch.distributed, between threads
        def __init__(self, r):
            self.r = r

        def barrier(self):
            bar.wait()

        def get_backend(self):
            return "threads"

        def all_reduce(self, t, op=None):
            out[self.r] = float(t[0])
            bar.wait()
            t[0] = max(out)
            bar.wait()

        def all_gather_object(self, lst, obj):
            out[self.r] = obj
            bar.wait()
            lst[:] = list(out)
            bar.wait()

        class ReduceOp:
            MAX = "max"

    def worker(r):
        try:
            if per * world <= len(cpus):
                os.sched_setaffinity(0, cpus[r * per:(r + 1) * per])  # (tid 0 = the calling thread)
            dev = r if rccl else 0

            def comm(ctx):
                if world == 1:
                    return None, None
                if rccl:
                    ctx.comm_init_rccl(uid, r, world)
                else:
                    ctx.comm_init_host(name, r, world)
                got = ctx.allreduce([r + 1.0, 1.0, -0.25 * (r + 1)])
                tot = 0.5 * world * (world + 1)
                if not np.array_equal(got, [tot, float(world), -0.25 * tot]):
                    raise RuntimeError(f"all-reduce self-check returned {got.tolist()}")
                return ctx.comm_info()["kind"], None

            class T:  # measure() only needs .tensor / .float64 for the max-over-ranks of one number
                float64 = "f64"

                @staticmethod
                def tensor(v, dtype=None, device=None):
                    class V(list):
                        def item(self):
                            return self[0]
                    return V(v)

                class cuda:
                    @staticmethod
                    def synchronize():
                        pass

            d = ThreadDist(r)
            res, ctx, model, _ = measure(capi, cfg, args.steps, args.warmup, r, world, dev, None, nthreads, comm, d, T)
            errs[r] = (res, ctx, model)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
            bar.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    bad = [e for e in errs if isinstance(e, BaseException)]
    if bad:
        first = [e for e in bad if not isinstance(e, threading.BrokenBarrierError)] or bad
        print(f"bench --inproc failed: {first[0]!r}", file=sys.stderr)
        return 1
    res = errs[0][0]
    res["config"]["parallelism"] += f"; ONE process, {world} host threads (one context per GPU)"
    line = {
        "metric": "E-step data-points/sec (full VBEM iteration: suff-stats + M-step + E-step)",
        "value": res["value"], "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "config": res["config"], "free_energy": res["free_energy"],
        "kernels": res["kernels"], "roofline": res["roofline"],
        **({"check": res["check"]} if "check" in res else {}),
    }
    print(json.dumps(line), flush=True)
    for _, ctx, model in errs:
        if model is not None:
            model.close()
        ctx.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="northstar", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--groups", type=int, default=0, help="override groups per GPU (grouped configurations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-matrix learnVDP call (needs ~9 GB of host memory)")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--parity-rows", type=int, default=1_000_000, help="N_par of SURVEY 8(d)")
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="multi-GPU all-reduce: the library's own RCCL collective, or the torch.distributed hook")
    ap.add_argument("--inproc", action="store_true",
                    help="one process drives the N GPUs from N host threads (the LIBCLUSTER_GPUS mode of the learners: "
                         "one context, stream and M-step pool per GPU, ncclCommInitRank per thread) instead of one "
                         "process per GPU")
    args = ap.parse_args()

    if args.inproc:
        raise SystemExit(main_inproc(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    narrowed = False
    if world > 1:
        # this rank's slice of the CPUs the job may use: its Python thread and the M-step pool the library starts (threads
        # inherit the mask) stay off the other ranks' cores
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // world
        if per >= 1:
            os.sched_setaffinity(0, cpus[rank * per:(rank + 1) * per])
            narrowed = True
    import torch

    if os.environ.get("LC_ALL_RANKS_ON_GPU0"):  # multi-rank smoke test on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # LC_BENCH_FORCE_DIST=1: walk the multi-rank code path (process group, RCCL unique-id broadcast, native communicator,
    # self-check, barriers) with a world of one -- the only way to exercise it on a one-GPU box
    force_dist = bool(os.environ.get("LC_BENCH_FORCE_DIST")) and "RANK" in os.environ
    if world > 1 or force_dist:
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
    if args.groups:
        cfg["J"] = args.groups
    # host M-step threads of this rank = the size of ITS mask: the mask above is already the rank's 1 / world slice (dividing
    # it by the world size again left one thread per rank on a 64-core node with 8 ranks); only when the ranks could not
    # be given disjoint slices (fewer CPUs than ranks) do they share the whole mask
    mine = len(os.sched_getaffinity(0))
    nthreads = max(1, min(32, mine if (narrowed or world == 1) else mine // world))
    nthreads = int(os.environ.get("LC_BENCH_THREADS", nthreads))
    stream = torch.cuda.current_stream().cuda_stream

    def comm(ctx):
        return setup_comm(ctx, capi, lcd, dist, torch, rank, world, local_rank, args.comm)

    res, ctx, model, (wkind, ckind, mu, L, family) = measure(capi, cfg, args.steps, args.warmup, rank, world,
                                                              local_rank, stream, nthreads, comm, dist, torch)
    if rank == 0:
        # HBM bytes per launch from the committed PMC passes (not live): the newest profiles/rNN_pmc_traffic.json, which
        # tools/pmc_traffic.py GENERATES from the rocprofv3 summaries next to it (tests/test_host.py checks that)
        tfs = sorted((ROOT / "profiles").glob("r[0-9][0-9]_pmc_traffic.json"))
        traffic = json.loads(tfs[-1].read_text()) if tfs else {}

        def put_traffic(roof, cfgname):
            e = traffic.get(cfgname, {})
            roof["traffic"] = e.get(roof["kernel"])
            lps = e.get("_launches_per_step", {}).get(roof["kernel"])
            if lps is not None:
                roof["traffic_launches_per_step"] = lps
            if roof["traffic"] is not None:
                roof["traffic_source"] = f"profiles/{tfs[-1].name} <- profiles/{e.get('_summary', '?')}"

        if not args.rows and not args.groups:
            put_traffic(res["roofline"], args.config)
        line = {
            "metric": "E-step data-points/sec (full VBEM iteration: suff-stats + M-step + E-step)",
            "value": res["value"], "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": res["config"], "free_energy": res["free_energy"],
            "kernels": res["kernels"], "roofline": res["roofline"], "host_phases_ms": res["host_phases_ms"],
            **({"check": res["check"]} if "check" in res else {}),
        }
        N, D, K = cfg["N"], cfg["D"], cfg["K"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline(ctx, model, cfg, args.cpu_sample_rows) if family == "GaussWish" else
                                    cpu_baseline_family(ctx, cfg, min(args.cpu_sample_rows, 400_000)))
        if world == 1 and not args.no_parity:
            if family == "GaussWish" and cfg.get("J", 1) == 1:
                sys.path.insert(0, str(ROOT / "oracle"))
                import lc_oracle_c as oc

                rows = min(args.parity_rows, N)
                nt = oc.physical_cores()
                line["parity"] = parity_c(capi, cfg, wkind, local_rank, rows, 3, False, nt)
                line["parity_overlap"] = parity_c(capi, cfg, wkind, local_rank, rows, 3, True, nt)
            else:
                line["parity"] = parity_numpy(capi, cfg, wkind, mu, L, local_rank, ckind=ckind)
        if world == 1 and args.config == "northstar" and not args.rows and not args.no_other_configs:
            # the other BASELINE configurations and the separable families, 5 timed steps each: driver-visible
            # ms_per_step / roofline of the kernels that are NOT on the headline line
            if model is not None:
                model.close()
            ctx.close()
            ctx = model = None
            others = []
            for name in OTHER_CONFIGS:
                c2 = dict(CONFIGS[name])
                # (the 0.18 ms iteration of the small configuration needs more steps for a steady number; so do the 4 ms
                #  iterations of the separable families: their first five read 5 % above a 20-step run)
                work = c2["N"] * c2["D"] * c2["K"]
                st, wu = (200, 20) if work < 1e9 else (20, 3) if (c2.get("c", "GaussWish") != "GaussWish" or work < 1e10) else (5, 1)
                r2, x2, m2, _ = measure(capi, c2, st, wu, 0, 1, local_rank, stream, nthreads, None, None, torch)
                put_traffic(r2["roofline"], name)
                others.append({"config": name, "workload": c2["label"], "steps": st, "warmup": wu,
                               "value": r2["value"], "ms_per_step": r2["ms_per_step"], "kernels": r2["kernels"],
                               "roofline": r2["roofline"], "host_phases_ms": r2["host_phases_ms"]})
                if m2 is not None:
                    m2.close()
                x2.close()
            line["other_configs"] = others
            line["model_selection"] = model_selection(capi, local_rank, stream)
            if not args.no_dropin:
                try:
                    line["dropin_call"] = dropin_call(capi, local_rank)
                except MemoryError as e:  # (a host with less than ~10 GB to spare: the line says so instead of dying)
                    line["dropin_call"] = {"skipped": f"not enough host memory for a 10M x 64 matrix: {e}"}
        line["summary"] = summary_of(line)  # LAST key: the driver's record keeps the tail of the line
        print(json.dumps(line), flush=True)
    if model is not None:
        model.close()
    if ctx is not None:
        ctx.close()  # (destroys the RCCL communicator while the process group is still up)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
