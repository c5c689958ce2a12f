"""Row-sharded model selection across ranks on the GPU box.  Runs FIRST (file name) and only
through child processes: this pytest process must not have touched the GPU when it starts them."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    out["_log"] = r.stdout + r.stderr
    return out


@pytest.mark.parametrize("mode,family,D", [("rows", "GaussWish", 6), ("groups", "GaussWish", 6), ("rows", "NormGamma", 6),
                                           ("groups", "ExpGamma", 6), ("rows", "ExpGamma", 6),
                                           # D = 24: cluster() on the journaled distance cache
                                           ("rows", "GaussWish", 24), ("groups", "GaussWish", 24)])
def test_two_rank_cluster_equals_single_rank(lib, mode, family, D):
    """cluster() (VBEM + prune + split search) sharded over two ranks -- by row blocks (BGMM) or by whole
    groups (GMC: per-group counts and weights stay local) -- with all-reduced statistics, Fz, LL_k and
    decision counts takes the same decisions and reaches the same F as one rank.  Also for the diagonal and the
    exponential family (whose split threshold is a global per-group mean in row-sharded runs)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = str(ROOT / "tools" / "dist_cluster_check.py")
    args = ["30000", str(D), "5", mode, family]
    one = _run([sys.executable, script, *args])
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), script, *args],
               {"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"})
    assert two["world"] == 2 and one["world"] == 1
    assert one["K"] == two["K"] and one["K"] >= (5 if family != "ExpGamma" else 2)
    assert [k for k, _ in one["rounds"]] == [k for k, _ in two["rounds"]]
    for (_, a), (_, b) in zip(one["rounds"], two["rounds"]):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    assert abs(one["F"] - two["F"]) <= 1e-10 * abs(one["F"])
    np.testing.assert_allclose(one["N"], two["N"], rtol=1e-9)


@pytest.mark.parametrize("mode", ["rows", "groups"])
def test_one_rank_failing_to_journal_a_cache_column_is_a_joint_fallback(lib, mode):
    """The distance cache's journal columns are reserved by every rank and the outcome is agreed on with one all-reduced
    flag: when ONE rank cannot reserve (LC_TEST_JOURNAL_FAIL_RANK: rank 1 pretends to be out of memory -- under group
    sharding ranks hold different numbers of rows), BOTH ranks leave the cache together and finish on the ordinary
    kernels; the collectives stay matched and the rounds, K and F are those of a single rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = str(ROOT / "tools" / "dist_cluster_check.py")
    args = ["30000", "24", "5", mode, "GaussWish"]
    one = _run([sys.executable, script, *args])
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), script, *args],
               {"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1", "LC_TEST_JOURNAL_FAIL_RANK": "1", "LC_TRACE_PHASES": "1",
                "LC_LIB_PATH": str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")})
    # both ranks gave the cache up together: rank 1 because its (test-hooked) reservation failed, rank 0 "on another rank"
    assert "LC_TEST_JOURNAL_FAIL_RANK" in two["_log"] and "on another rank" in two["_log"]
    assert two["world"] == 2 and one["K"] == two["K"] >= 5
    assert [k for k, _ in one["rounds"]] == [k for k, _ in two["rounds"]]
    for (_, a), (_, b) in zip(one["rounds"], two["rounds"]):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    assert abs(one["F"] - two["F"]) <= 1e-10 * abs(one["F"])


@pytest.mark.parametrize("model", ["scm", "mcm"])
def test_two_rank_topic_model_equals_single_rank(lib, model):
    """learnSCM / learnMCM with whole groups (and their documents) per rank: all-reduced cluster statistics, N_tk,
    document-level Gaussian statistics, Fyz / Fz and split-search counts reproduce the single-rank rounds and F."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = str(ROOT / "tools" / "dist_topic_check.py")
    one = _run([sys.executable, script, model])
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), script, model],
               {"LC_DIST_BACKEND": "gloo", "LC_ALL_RANKS_ON_GPU0": "1"})
    assert two["world"] == 2 and one["world"] == 1
    assert (one["T"], one["K"]) == (two["T"], two["K"]) and one["K"] >= 3
    assert [(t, k) for t, k, _ in one["rounds"]] == [(t, k) for t, k, _ in two["rounds"]]
    for (_, _, a), (_, _, b) in zip(one["rounds"], two["rounds"]):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    assert abs(one["F"] - two["F"]) <= 1e-10 * abs(one["F"])
    np.testing.assert_allclose(one["N"], two["N"], rtol=1e-9)
