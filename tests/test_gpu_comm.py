"""The library's own collectives (libcluster_amd/csrc/lc_comm.cpp) and the one-call multi-GPU switch.

  * tests/cpp/dist_test.cpp: two ranks = two processes through the C ABI alone (no Python, no torch) reproduce the
    single-rank model selection -- host-staged transport on one GPU, RCCL when the box has two GPUs (world of one
    otherwise, the two-rank part reported as skipped);
  * LIBCLUSTER_GPUS: learnBGMM / learnVDP / learnGMC / ... shard their observations over several contexts inside one
    call (one host thread per shard) and return F, rounds, qZ, weights and clusters as from one GPU
    (LIBCLUSTER_GPUS_SAME_DEVICE=1 puts every shard on GPU 0 with the host-staged transport);
  * the RCCL communicator with a world of one inside the EM loop.
The loop being distributed: src/cluster.cpp:207-223 (single-process OpenMP in the reference)."""
import os
import signal
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "tests" / "cpp" / "_build" / "dist_test"


def _compile():
    from libcluster_amd import capi

    EXE.parent.mkdir(exist_ok=True)
    libdir = capi.LIB_PATH.parent
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", f"-I{ROOT / 'include'}", str(ROOT / "tests/cpp/dist_test.cpp"),
           "-o", str(EXE), f"-L{libdir}", "-lcluster_hip", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_dist_test_compiles_against_the_c_abi(lib):
    _compile()


def test_comm_symbols_and_lazy_rccl(lib):
    """The collectives are part of the C ABI; librccl is bound on demand, not at load time."""
    from libcluster_amd import capi

    for s in ("lc_comm_unique_id", "lc_ctx_comm_init_rccl", "lc_ctx_comm_init_host", "lc_ctx_comm_free",
              "lc_ctx_comm_info", "lc_ctx_allreduce", "lc_comm_rccl_available", "lc_source_hash"):
        assert hasattr(lib, s), s
    out = subprocess.run(["ldd", str(capi.LIB_PATH)], capture_output=True, text=True).stdout
    assert "librccl" not in out  # dlopen on first use: single-GPU callers never page in the 570 MB library


def _run_group(cmd, timeout):
    """Run in its own process group so that a hung rank is killed with its parent."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True,
                         cwd=str(ROOT))
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, _ = p.communicate()
        pytest.fail("dist_test timed out:\n" + out[-3000:])
    return p.returncode, out


@pytest.mark.gpu
@pytest.mark.parametrize("mode,D", [("host", 8), ("rccl", 8), ("host", 24)])  # (D = 24: on the distance cache)
def test_two_ranks_through_the_c_abi(lib, mode, D):
    exe = _compile()
    rc, out = _run_group([str(exe), mode, "40000", str(D), "4"], 300)
    assert rc == 0, out[-3000:]
    assert f"dist_test {mode} OK" in out, out[-3000:]
    if mode == "host":
        assert "2 ranks" in out and "host all-reduce" in out


def _blobs(seed, n, D, K, sep=6.0):
    rng = np.random.default_rng(seed)
    mu = rng.normal(0, sep, (K, D))
    z = rng.integers(0, K, n)
    return mu[z] + rng.normal(size=(n, D)) * rng.uniform(0.5, 1.2, (K, 1))[z]


@pytest.fixture
def sharded_env():
    keys = ("LIBCLUSTER_GPUS", "LIBCLUSTER_GPUS_SAME_DEVICE", "LIBCLUSTER_COMM")
    old = {k: os.environ.get(k) for k in keys}

    def set_(n):
        if n:
            os.environ["LIBCLUSTER_GPUS"] = str(n)
            os.environ["LIBCLUSTER_GPUS_SAME_DEVICE"] = "1"
        else:
            for k in keys:
                os.environ.pop(k, None)

    yield set_
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("learner,shards", [("learnBGMM", 2), ("learnVDP", 3), ("learnDGMM", 2), ("learnBEMM", 2)])
def test_libcluster_gpus_rows_sharded_equals_one_gpu(lib, sharded_env, learner, shards):
    """LIBCLUSTER_GPUS inside the single-matrix learners: row blocks on `shards` contexts, statistics summed by the
    library's collective, replicated M-step -- the same rounds, K, F, qZ, weights and clusters as one context."""
    import libcluster_amd as lc

    X = _blobs(5, 6001, 5, 4)
    if learner == "learnBEMM":
        X = np.abs(X) * np.array([1.0, 10.0, 100.0, 1.0, 5.0])
    fn = getattr(lc, learner)
    sharded_env(0)
    F1, q1, w1, *rest1, info1 = fn(X, return_info=True)
    sharded_env(shards)
    F2, q2, w2, *rest2, info2 = fn(X, return_info=True)
    assert info1["K"] == info2["K"] and info1["K"] >= 2
    assert [k for k, _ in info1["rounds"]] == [k for k, _ in info2["rounds"]]
    assert abs(F1 - F2) <= 1e-10 * abs(F1)
    assert q2.shape == q1.shape
    np.testing.assert_allclose(q2, q1, atol=1e-9)
    np.testing.assert_allclose(w2, w1, rtol=1e-9)
    for a, b in zip(rest1, rest2):
        for u, v in zip(a, b):
            if u is None:  # exponential clusters have no covariance
                assert v is None
            else:
                np.testing.assert_allclose(v, u, rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("learner,sparse", [("learnGMC", False), ("learnSGMC", True), ("learnGMC", True), ("learnDGMC", False)])
def test_libcluster_gpus_whole_groups_equals_one_gpu(lib, sharded_env, learner, sparse):
    """GMC family under LIBCLUSTER_GPUS: whole groups per shard (largest first), per-group counts and weights stay with
    the shard that holds the group, qZ and weights come back in the caller's group order."""
    import libcluster_amd as lc

    if sparse:  # two alternating, well separated mixtures: groups without mass in half of the clusters (the sparse
        # updates are an approximation -- on less separated data the reference itself stops with "Free energy increase!")
        sizes = [300, 500, 200, 400, 250]
        X = [_blobs(100 + (j % 2), n, 3, 3, 10.0) for j, n in enumerate(sizes)]
    else:
        sizes = [900, 1500, 400, 1200, 700]
        X = [_blobs(40 + j, n, 4, 3 + (j % 2)) + (j % 2) * 3.0 for j, n in enumerate(sizes)]
    fn = getattr(lc, learner)
    sharded_env(0)
    F1, q1, w1, *rest1, info1 = fn(X, sparse=sparse, return_info=True)
    sharded_env(3)
    F2, q2, w2, *rest2, info2 = fn(X, sparse=sparse, return_info=True)
    assert info1["K"] == info2["K"] and info1["K"] >= 2
    assert [k for k, _ in info1["rounds"]] == [k for k, _ in info2["rounds"]]
    assert abs(F1 - F2) <= 1e-10 * abs(F1)
    assert len(q2) == len(sizes)
    for a, b in zip(q1, q2):
        assert a.shape == b.shape
        np.testing.assert_allclose(b, a, atol=1e-9)
    for a, b in zip(w1, w2):
        np.testing.assert_allclose(b, a, rtol=1e-9)


@pytest.mark.gpu
def test_rccl_world_of_one_inside_the_em_loop(lib):
    """ncclCommInitRank + ncclAllReduce(ncclDouble, ncclSum) on the context's stream in every iteration: with one rank
    the sums are identities, so the trace must equal the communicator-free run bit for bit."""
    from libcluster_amd import capi

    if not capi.rccl_available():
        pytest.skip("librccl could not be loaded on this box")
    X = _blobs(9, 20000, 16, 5)
    q0 = np.random.default_rng(1).dirichlet(np.ones(5), X.shape[0])
    traces = []
    for use in (False, True):
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q0)
            if use:
                ctx.comm_init_rccl(capi.comm_unique_id(), 0, 1)
                info = ctx.comm_info()
                assert info == {"rank": 0, "world": 1, "kind": "rccl"}
                np.testing.assert_array_equal(ctx.allreduce([1.5, -2.0, 3.25]), [1.5, -2.0, 3.25])
            F, tr, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=4)
            m.close()
            if use:
                ctx.comm_free()
                assert ctx.comm_info()["kind"] == "none"
        traces.append(tr)
    np.testing.assert_array_equal(traces[0], traces[1])


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [0, 2])
def test_per_group_weight_priors_reach_the_learner(lib, sharded_env, shards):
    """learnSGMC with Dirichlet(alpha_j) objects in the caller's `weights` vector: vbem's weights.resize(J, W()) keeps
    them (cluster.cpp:192), so group j learns with ITS alpha.  lc_learn_w carries the priors; equal to the oracle's
    cluster() with the same weight objects, unsharded and sharded over whole groups."""
    import lc_oracle as o
    from libcluster_amd import capi

    sizes = [300, 500, 200, 400]
    X = [_blobs(100 + (j % 2), n, 3, 3, 10.0) for j, n in enumerate(sizes)]
    alphas = [0.1, 2.5, 1.0, 0.4]
    w = [o.Dirichlet(a) for a in alphas]
    cl = []
    Fo, qo = o.cluster(X, w, cl, 1.0, -1, False, False, o.Dirichlet)
    sharded_env(shards)
    F, m, rows = capi.learn(capi.ALGO_SGMC, X, 1.0, 1.0, -1, False, False, 4, 0, wprior_j=alphas)
    q = m.qz_all(rows)
    K = m.dims()[1]
    el = [m.weights(j)[0] for j in range(len(X))]
    m.close()
    assert K == len(cl)
    assert abs(F - Fo) <= 1e-9 * abs(Fo)
    for a, b in zip(q, qo):
        np.testing.assert_allclose(a, b, atol=1e-8)
    for j in range(len(X)):
        np.testing.assert_allclose(el[j], w[j].Elogweight(), rtol=1e-8, atol=1e-10)
    # and they matter: the default priors give another free energy
    sharded_env(0)
    F1, m1, _ = capi.learn(capi.ALGO_SGMC, X, 1.0, 1.0, -1, False, False, 4, 0)
    m1.close()
    assert abs(F1 - F) > 1e-6 * abs(F)


@pytest.mark.gpu
def test_libcluster_gpus_rccl_init_all_with_one_device(lib):
    """The in-process multi-GPU path with its RCCL transport (ncclCommInitAll, one host thread per shard, all-reduce
    on the shard's own stream) -- on a one-GPU box with a world of one (LIBCLUSTER_FORCE_SHARDED lets a single shard
    take the sharded path): equal to the ordinary call.  More than one RCCL rank needs more than one GPU."""
    import json
    import subprocess
    import sys

    code = (
        "import numpy as np, json, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import libcluster_amd as lc\n"
        "rng = np.random.default_rng(5)\n"
        "mu = rng.normal(0, 6.0, (4, 5)); z = rng.integers(0, 4, 6001)\n"
        "X = mu[z] + rng.normal(size=(6001, 5))\n"
        "F, q, w, m, c, info = lc.learnVDP(X, return_info=True)\n"
        "print('RESULT ' + json.dumps({'F': F, 'K': info['K'], 'rounds': [k for k, _ in info['rounds']], 'qsum': float(q.sum()),"
        " 'q0': q[:5].tolist()}))\n")
    outs = []
    for env in ({}, {"LIBCLUSTER_GPUS": "1", "LIBCLUSTER_FORCE_SHARDED": "1"}):
        e = dict(os.environ)
        for k in ("LIBCLUSTER_GPUS", "LIBCLUSTER_GPUS_SAME_DEVICE", "LIBCLUSTER_FORCE_SHARDED", "LIBCLUSTER_COMM"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads([x for x in r.stdout.splitlines() if x.startswith("RESULT ")][-1][7:]))
    a, b = outs
    assert a["K"] == b["K"] and a["rounds"] == b["rounds"]
    assert abs(a["F"] - b["F"]) <= 1e-12 * abs(a["F"])
    np.testing.assert_allclose(a["q0"], b["q0"], atol=1e-12)


@pytest.mark.gpu
def test_bench_multi_rank_path_with_a_world_of_one(lib):
    """bench.py's N > 1 path end to end -- torch.distributed process group on RCCL, broadcast of the unique id,
    lc_ctx_comm_init_rccl, the self-check sum, barriers, max-over-ranks -- with one rank (LC_BENCH_FORCE_DIST): the line
    names the library's own collective."""
    import json
    import socket
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    e["LC_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1",
                        "--config", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective"] == "rccl"
    assert line["value"] > 0 and np.isfinite(line["free_energy"])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(lib):
    """`python bench.py --gpus 2` without a launcher: two child ranks (both on GPU 0 here, gloo rendezvous, the library's
    host-staged collective), rank 0's line on stdout, exit status 0."""
    import json
    import sys

    e = dict(os.environ)
    e.update({"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"})
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "host-shm"
    assert line["config"]["rows_per_gpu"] == 200_000 and line["value"] > 0
    _check_per_rank(line, 2)
    ncpu = len(os.sched_getaffinity(0))
    for r in line["config"]["per_rank"]:  # the M-step pool covers the rank's whole slice of the CPUs
        assert r["cpus"] == (ncpu // 2 if ncpu >= 2 else ncpu) and r["mstep_threads"] == min(32, r["cpus"]), r


def _bench_line(args, env, timeout=1500):
    import json
    import sys

    e = dict(os.environ)
    e.update(env)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args, "--no-cpu-baseline", "--no-parity", "--no-other-configs"],
                       capture_output=True, text=True, timeout=timeout, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("config,one,eight", [
    # BASELINE config 4's shape (BGMM, D = 64, K = 32), 8 row blocks of 1.25M rows = ranks 0..7 of the one Philox stream
    ("northstar", ["--rows", "10000000"], ["--rows", "1250000"]),
    # BASELINE config 5's shape (GMC, D = 128, K = 64): 64 groups of 60k rows, eight whole groups per rank
    ("5", ["--rows", "3840000", "--groups", "64"], ["--rows", "480000", "--groups", "8"]),
])
def test_bench_eight_ranks_on_one_gpu_equal_one_rank(lib, config, one, eight):
    """The `--gpus 8` path the driver's scaling run takes, executed with a world of EIGHT before it ever meets eight GPUs:
    eight processes (torch.distributed.run started by bench.py itself, gloo rendezvous, all on GPU 0), the library's
    native host-staged collective in every iteration, the eight-entry per-rank report -- and the same model as ONE rank
    over the same rows: free energy and what the M-step made of the reduced statistics (N_k, means, scatter traces,
    log-determinants) to 1e-11.  Also: every rank's M-step runs on ALL the CPUs of its slice (the affinity mask used to
    be divided by the world size twice).  The loop being distributed: src/cluster.cpp:207-223."""
    base = ["--config", config, "--steps", "3", "--warmup", "1"]
    a = _bench_line(["--gpus", "1", *base, *one], {})
    b = _bench_line(["--gpus", "8", *base, *eight], {"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"})
    assert a["n_gpus"] == 1 and b["n_gpus"] == 8
    assert b["config"]["collective"] == "host-shm"  # the library's own transport, not the torch.distributed hook
    assert len(b["config"]["per_rank"]) == 8
    _check_per_rank(b, 8)
    ncpu = len(os.sched_getaffinity(0))
    for r in b["config"]["per_rank"]:
        assert r["mstep_threads"] == min(32, r["cpus"]) or ncpu < 8, r
        assert r["cpus"] == (ncpu // 8 if ncpu >= 8 else ncpu), r
    assert abs(a["free_energy"] - b["free_energy"]) <= 1e-11 * abs(a["free_energy"])
    for key, tol in (("Nk", 1e-11), ("mean_sum", 1e-9), ("iW_trace", 1e-11), ("logdW", 1e-11)):
        x, y = np.array(a["check"][key]), np.array(b["check"][key])
        assert x.shape == y.shape and x.size == a["config"]["K"]
        np.testing.assert_allclose(y, x, rtol=tol, atol=tol * (1.0 + np.max(np.abs(x))), err_msg=key)


@pytest.mark.gpu
def test_bench_eight_inproc_shards_equal_eight_processes(lib):
    """`--gpus 8 --inproc` (one process, eight host threads / contexts / streams / M-step pools: the LIBCLUSTER_GPUS=8
    deployment) against eight processes on the same rows: same free energy, eight per-rank entries."""
    base = ["--gpus", "8", "--config", "northstar", "--rows", "400000", "--steps", "3", "--warmup", "1"]
    env = {"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"}
    a = _bench_line([*base, "--inproc"], env)
    b = _bench_line(base, env)
    assert a["config"]["collective"] == "host-shm" == b["config"]["collective"] and "ONE process" in a["config"]["parallelism"]
    assert len(a["config"]["per_rank"]) == 8
    _check_per_rank(a, 8)
    assert abs(a["free_energy"] - b["free_energy"]) <= 1e-12 * abs(b["free_energy"])
    np.testing.assert_allclose(a["check"]["Nk"], b["check"]["Nk"], rtol=1e-12)


def _check_per_rank(line, world):
    """The per-rank report a first real multi-GPU run is diagnosed with: every rank's kernel times, its exchange step
    (events around the collective), host M-step and residual wait, consistent with the step time."""
    pr = line["config"]["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world))
    for r in pr:
        for key in ("step_ms", "estep_ms", "suffstat_ms", "allreduce_ms", "mstep_ms", "wait_ms", "mstep_threads", "cpus"):
            assert key in r and np.isfinite(r[key]) and r[key] >= 0, (key, r)
        assert r["allreduce_calls_per_step"] >= 1          # statistics (+ F_z) are summed over ranks every iteration
        assert r["allreduce_ms"] > 0 and r["mstep_ms"] > 0
        assert r["step_ms"] <= line["ms_per_step"] * 1.001 + 1e-6  # the line carries the MAX over ranks
        busy = r.get("fused_ms", 0.0) + r["estep_ms"] + r["suffstat_ms"]
        assert busy + r["mstep_ms"] <= 1.5 * r["step_ms"] + 1.0


@pytest.mark.gpu
def test_bench_inproc_threads_drive_the_shards(lib):
    """`bench.py --gpus 2 --inproc`: ONE process, one host thread + context per shard (the LIBCLUSTER_GPUS mode), the
    library's collective between them (host-staged here: both shards on GPU 0); the same free energy as the
    process-per-GPU mode on the same rows."""
    import json
    import sys

    e = dict(os.environ)
    e.update({"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"})
    e.pop("WORLD_SIZE", None)
    lines = []
    for extra in (["--inproc"], []):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "3",
                            "--warmup", "1", *extra], capture_output=True, text=True, timeout=900, env=e, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines.append(json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1]))
    a, b = lines
    assert a["n_gpus"] == 2 and a["config"]["collective"] == "host-shm" and "ONE process" in a["config"]["parallelism"]
    _check_per_rank(a, 2)
    assert abs(a["free_energy"] - b["free_energy"]) <= 1e-12 * abs(b["free_energy"])


@pytest.mark.gpu
@pytest.mark.parametrize("learner,shards,sep", [("learnVDP", 3, 6.0), ("learnBGMM", 2, 0.6), ("learnGMC", 3, 5.0)])
def test_sharded_model_selection_on_cached_distances_equals_one_gpu(lib, sharded_env, learner, shards, sep):
    """D = 20: cluster() runs on the journaled distance cache with statistics that follow the moved rows.
    Sharded over contexts, every decision of that machinery (which columns to recompute, whether the moved rows are few
    enough, whether the cache still pays, whether there is room) must come out the same on every shard: same rounds, K,
    F and responsibilities as one context, for separated and for overlapping clusters."""
    import libcluster_amd as lc

    if learner == "learnGMC":
        X = [_blobs(60, n, 20, 5, sep) for n in (1500, 900, 2100, 600)]
    else:
        X = _blobs(61, 7001, 20, 5, sep)
    fn = getattr(lc, learner)
    sharded_env(0)
    F1, q1, w1, *rest1, info1 = fn(X, return_info=True)
    sharded_env(shards)
    F2, q2, w2, *rest2, info2 = fn(X, return_info=True)
    assert info1["K"] == info2["K"] and info1["K"] >= 2
    assert [k for k, _ in info1["rounds"]] == [k for k, _ in info2["rounds"]]
    for (_, a), (_, b) in zip(info1["rounds"], info2["rounds"]):
        np.testing.assert_allclose(b, a, rtol=1e-10)
    assert abs(F1 - F2) <= 1e-10 * abs(F1)
    if learner == "learnGMC":
        for a, b in zip(q1, q2):
            np.testing.assert_allclose(b, a, atol=1e-9)
    else:
        np.testing.assert_allclose(q2, q1, atol=1e-9)


_GATHER_SNIPPET = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
from libcluster_amd import capi
rng = np.random.default_rng(9)
mu = rng.normal(0, 6.0, (5, 16))
X = mu[rng.integers(0, 5, 20000)] + rng.normal(size=(20000, 16))
q0 = np.random.default_rng(1).dirichlet(np.ones(5), X.shape[0])
with capi.Context(0) as ctx:
    ctx.set_data(X)
    ctx.set_qz(q0)
    ctx.comm_init_rccl(capi.comm_unique_id(), 0, 1)
    print("KIND", ctx.comm_info()["kind"])
    v = ctx.allreduce([1.5, -2.0, 3.25])
    assert list(v) == [1.5, -2.0, 3.25], v
    big = rng.normal(size=300000)          # (the gather buffer grows between calls)
    assert np.array_equal(ctx.allreduce(big), big)
    F, tr, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=4)
    m.close()
    ctx.comm_free()
print("TRACE", " ".join(float(f).hex() for f in tr))
"""


@pytest.mark.gpu
def test_rccl_gather_mode_world_of_one_inside_the_em_loop(lib):
    """LIBCLUSTER_COMM=rccl-gather: ncclAllGather + the rank-order sum kernel instead of ncclAllReduce (lc_comm.cpp), so
    that the sum over ranks does not depend on RCCL's ring order.  With one rank it is an identity: the trace of a VBEM
    run must equal the default transport's bit for bit.  (More than one RCCL rank needs more than one GPU; the order of
    the additions is tested separately below, against the host transport's loop.)"""
    from libcluster_amd import capi

    if not capi.rccl_available():
        pytest.skip("librccl could not be loaded on this box")
    outs = {}
    for mode in ("", "rccl-gather"):
        e = dict(os.environ)
        e.pop("LIBCLUSTER_COMM", None)
        if mode:
            e["LIBCLUSTER_COMM"] = mode
        r = subprocess.run([sys.executable, "-c", _GATHER_SNIPPET.format(root=str(ROOT))], capture_output=True, text=True,
                           timeout=600, env=e, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = r.stdout
    assert "KIND rccl\n" in outs[""] and "KIND rccl-gather\n" in outs["rccl-gather"]
    tr = {k: [ln for ln in v.splitlines() if ln.startswith("TRACE ")][-1] for k, v in outs.items()}
    assert tr[""] == tr["rccl-gather"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,count", [(1, 7), (2, 1000), (8, 4097), (8, 1100000), (5, 263)])
def test_rank_order_sum_kernel_adds_like_the_host_transport(lib, world, count):
    """The device half of LIBCLUSTER_COMM=rccl-gather (rank_order_sum_kernel) against the additions of the host
    transport (HostComm::allreduce_sum: slot 0, then += slot 1, 2, ... -- tests/test_gpu_comm.py's 8-rank runs pin that
    path to one rank): values that differ by many orders of magnitude, so that any other order of the additions, or a
    fused multiply-add, changes the low bits."""
    import ctypes as C

    hooked = C.CDLL(str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so"))
    fn = hooked.lc_test_rank_order_sum
    fn.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
    fn.restype = C.c_int
    rng = np.random.default_rng(world * 1000 + count)
    slots = rng.normal(size=(world, count)) * 10.0 ** rng.integers(-12, 12, size=(world, count))
    slots = np.ascontiguousarray(slots)
    out = np.empty(count)
    assert fn(slots.ctypes.data, world, count, out.ctypes.data) == 0
    want = slots[0].copy()
    for r in range(1, world):
        want += slots[r]
    assert np.array_equal(out, want)
    if world > 2:  # (the check can tell orders apart on this input)
        assert not np.array_equal(out, slots[::-1].cumsum(axis=0)[-1]) or count < 64
