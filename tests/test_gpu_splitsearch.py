"""The split search's shortcuts against the reference's literal schedule.

`split_gr` (cluster.cpp:366-495) runs a full `vbem(..., 1)` on all data for every candidate.  The HIP path keeps the
arithmetic and shares what is provably common between E-steps: LL_k as a by-product of the last E-step (no extra
pass), the statistics of the unchanged columns (a candidate's moved mass only), the distances of every cluster whose
posterior has not changed in any bit (a journaled cache, rolled back when a candidate is rejected), statistics that
follow the rows an E-step moved.  Each shortcut has an environment switch that restores the literal schedule; the two
must walk the same rounds to the same K and F, and both must equal the oracle."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu
HOOKED = str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")


def _with_switches(env):
    """The schedule switches (LC_SPLIT_*, LC_LL_EXTRA_PASS, LC_FUSED_SMALL, ...) exist in the test-hooks build of the library
    only (lck::test_switch): a run that sets one loads that build."""
    e = dict(env)
    if any(k.startswith(("LC_SPLIT_", "LC_LL_", "LC_FUSED_", "LC_SS_", "LC_ED_", "LC_ES_")) for k in e):
        e.setdefault("LC_LIB_PATH", HOOKED)
    return e



def _learn(args, env):
    e = dict(os.environ)
    e.update(_with_switches(env))
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "learn_bench.py"), *args], capture_output=True, text=True,
                       timeout=600, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"found K=(\d+) F=([-0-9.e+]+) in .*?; (\d+) rounds, (\d+) main VBEM iterations", r.stdout)
    assert m, r.stdout
    launches = re.search(r"E-step launches (\d+) .*suff-stat launches (\d+)", r.stdout)
    return int(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4)), int(launches.group(1))


LITERAL = {"LC_SPLIT_FULL_STATS": "1", "LC_SPLIT_NO_DCACHE": "1", "LC_LL_EXTRA_PASS": "1", "LC_FUSED_SMALL": "0"}


@pytest.mark.parametrize("args", [("300000", "24", "9"), ("200000", "64", "6", "Dirichlet"),
                                  ("400000", "12", "7"), ("250000", "40", "8", "Dirichlet", "NormGamma")])
def test_shortcuts_walk_the_same_rounds_as_the_literal_schedule(lib, args):
    fast = _learn(list(args), {})
    lit = _learn(list(args), LITERAL)
    assert fast[0] == lit[0] and fast[2] == lit[2] and fast[3] == lit[3], (fast, lit)
    assert abs(fast[1] - lit[1]) <= 1e-11 * abs(lit[1]), (fast, lit)
    assert fast[0] >= int(args[2]) - 1  # the final round tries (and rejects) every cluster: many candidates per round


def _learn_trace(args, env):
    import json

    e = dict(os.environ)
    e.update(_with_switches(env))
    e["LC_LB_TRACE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "learn_bench.py"), *args], capture_output=True, text=True,
                       timeout=900, env=e, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("TRACE ")][-1]
    return json.loads(line[6:])


@pytest.mark.parametrize("args,scale", [(("2000000", "64", "10"), "4.0"),       # well separated: the cache carries the loop
                                        (("1500000", "32", "8", "Dirichlet"), "1.0")])  # overlapping: rows keep moving
def test_mid_size_model_selection_cache_against_full_passes(lib, args, scale):
    """The sizes at which the distance cache and the moved-row statistics matter (millions of rows; the oracle cannot
    follow there): the default policy (tau = 2^-50, chained moved-row updates) against LC_SPLIT_NO_DELTA=1 (full
    E-step and statistics passes everywhere, the reference's schedule, cluster.cpp:473, 594-606) -- the same rounds,
    the same K per round, the same number of VBEM iterations per round and EVERY free energy of every round to 1e-10."""
    fast = _learn_trace(list(args), {"LC_LB_SCALE": scale})
    full = _learn_trace(list(args), {"LC_LB_SCALE": scale, "LC_SPLIT_NO_DELTA": "1"})
    assert [k for k, _ in fast] == [k for k, _ in full]
    assert fast[-1][0] >= int(args[2]) - 1
    for (k, a), (_, b) in zip(fast, full):
        assert len(a) == len(b), (k, a, b)
        np.testing.assert_allclose(a, b, rtol=1e-10, err_msg=f"round with K={k}")


def test_cached_first_estep_equals_the_oracle_on_a_many_candidate_round(lib):
    """Model selection on data whose final round rejects every candidate (K candidates, the cached path from the third
    on), every round's K and F against the oracle."""
    import lc_oracle as o
    import libcluster_amd as lc

    rng = np.random.default_rng(77)
    K, D, N = 6, 20, 9000
    mu = rng.normal(0, 5.0, (K, D))
    X = mu[rng.integers(0, K, N)] + rng.normal(size=(N, D))
    tr = []
    Fo, qo, _, clo = o.learnBGMM(X, trace=tr)
    F, q, w, means, covs, info = lc.learnBGMM(X, return_info=True)
    assert info["K"] == len(clo) >= K - 1
    assert [k for k, _ in info["rounds"]] == [k for k, _ in tr]
    for (_, a), (_, b) in zip(info["rounds"], tr):
        np.testing.assert_allclose(a, b, rtol=1e-9)
    assert abs(F - Fo) <= 1e-9 * abs(Fo)
    np.testing.assert_allclose(q, qo, atol=1e-8)


_SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
import libcluster_amd as lc
rng = np.random.default_rng({seed})
K, D, N, J, scale = {K}, {D}, {N}, {J}, {scale}
mu = rng.normal(0, scale, (K, D))
Xs = [mu[rng.integers(0, K, N)] + rng.normal(size=(N, D)) for _ in range(J)]
if J == 1:
    F, q, w, means, covs, info = lc.learnVDP(Xs[0], return_info=True)
else:
    F, q, w, means, covs, info = lc.learnGMC(Xs, return_info=True)
import hashlib
qs = [q] if J == 1 else list(q)
sha = hashlib.sha256(b"".join(np.ascontiguousarray(a, dtype=np.float64).tobytes() for a in qs)).hexdigest()
print(json.dumps(dict(F=F, K=info["K"], rounds=[[k, list(map(float, t))] for k, t in info["rounds"]], qsha=sha,
                      Fhex=float(F).hex())))
"""


def _run_snippet(env, **kw):
    import json

    e = dict(os.environ)
    e.update(_with_switches(env))
    r = subprocess.run([sys.executable, "-c", _SNIPPET.format(root=str(ROOT), **kw)], capture_output=True, text=True,
                       timeout=600, env=e, cwd=str(ROOT))
    if env.get("_EXPECT_FAILURE"):
        return dict(rc=r.returncode, _stderr=r.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["_stderr"] = r.stderr
    return out


@pytest.mark.parametrize("kw", [dict(seed=5, K=5, D=20, N=6000, J=1, scale=0.8),    # heavy overlap: the cache stops paying
                                dict(seed=6, K=6, D=24, N=5000, J=1, scale=2.0),    # some overlap
                                dict(seed=7, K=7, D=33, N=4000, J=1, scale=6.0),    # well separated
                                dict(seed=8, K=5, D=20, N=1500, J=3, scale=3.0)])   # groups (learnGMC)
def test_cached_distances_and_moved_row_statistics_against_full_passes_and_the_oracle(lib, kw):
    """cluster() on the journaled distance cache (Context::estep_cache) with statistics that follow the moved rows
    (Context::delta_suffstat): the default policy, the cache forced on however much moves, a zero tolerance (every row
    that moved at all counts) and full passes everywhere must walk the same rounds; the default must equal the oracle."""
    import lc_oracle as o

    runs = {name: _run_snippet(env, **kw) for name, env in
            {"default": {}, "forced": {"LC_SPLIT_DELTA_FORCE": "1"},
             "forced_tol0": {"LC_SPLIT_DELTA_FORCE": "1", "LC_SPLIT_DELTA_TOL": "0"},
             "full": {"LC_SPLIT_NO_DELTA": "1"}}.items()}
    ref = runs["full"]
    for name, r in runs.items():
        assert r["K"] == ref["K"] and [k for k, _ in r["rounds"]] == [k for k, _ in ref["rounds"]], (name, r, ref)
        for (_, a), (_, b) in zip(r["rounds"], ref["rounds"]):
            np.testing.assert_allclose(a, b, rtol=1e-10, err_msg=name)
        assert abs(r["F"] - ref["F"]) <= 1e-10 * abs(ref["F"]), name
    rng = np.random.default_rng(kw["seed"])
    mu = rng.normal(0, kw["scale"], (kw["K"], kw["D"]))
    Xs = [mu[rng.integers(0, kw["K"], kw["N"])] + rng.normal(size=(kw["N"], kw["D"])) for _ in range(kw["J"])]
    tr = []
    if kw["J"] == 1:
        Fo, _, _, clo = o.learnVDP(Xs[0], trace=tr)
    else:
        Fo, _, _, clo = o.learnGMC(Xs, trace=tr)
    d = runs["default"]
    assert d["K"] == len(clo) and [k for k, _ in d["rounds"]] == [k for k, _ in tr]
    for (_, a), (_, b) in zip(d["rounds"], tr):
        np.testing.assert_allclose(a, b, rtol=1e-9)
    assert abs(d["F"] - Fo) <= 1e-9 * abs(Fo)


def test_model_selection_carries_on_when_the_distance_cache_does_not_fit(lib):
    """LC_TEST_CACHE_NO_ROOM=K pretends the device is full from K clusters on: cluster() must finish on the ordinary
    kernels with the same rounds, K and F."""
    kw = dict(seed=7, K=7, D=33, N=4000, J=1, scale=6.0)
    a = _run_snippet({}, **kw)
    hooked = str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
    b = _run_snippet({"LC_TEST_CACHE_NO_ROOM": "4", "LC_LIB_PATH": hooked, "LC_TRACE_PHASES": "1"}, **kw)
    assert "distance cache given up" in b["_stderr"] and "no room for the distance cache at K = 4" in b["_stderr"]
    # the shipped library has no fault hooks: the switch does nothing there
    c = _run_snippet({"LC_TEST_CACHE_NO_ROOM": "4", "LC_TRACE_PHASES": "1"}, **kw)
    assert "distance cache given up" not in c["_stderr"] and c["K"] == a["K"]
    # ... and no schedule switches either (lck::test_switch): the other schedule's launch count shows only in the hooks build
    d = _learn(["60000", "24", "5"], {"LC_SPLIT_NO_DCACHE": "1", "LC_LIB_PATH": str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip.so")})
    e = _learn(["60000", "24", "5"], {})
    f = _learn(["60000", "24", "5"], {"LC_SPLIT_NO_DCACHE": "1"})
    assert d[4] == e[4] and f[4] != e[4] and d[:4] == e[:4] == f[:4], (d, e, f)
    assert a["K"] == b["K"] >= 6 and [k for k, _ in a["rounds"]] == [k for k, _ in b["rounds"]]
    for (_, x), (_, y) in zip(a["rounds"], b["rounds"]):
        np.testing.assert_allclose(x, y, rtol=1e-10)


_FP_CASES = [dict(seed=6, K=6, D=24, N=5000, J=1, scale=2.0), dict(seed=7, K=7, D=33, N=4000, J=1, scale=6.0),
             dict(seed=8, K=5, D=20, N=1500, J=3, scale=3.0), dict(seed=9, K=9, D=40, N=30000, J=1, scale=1.2)]


@pytest.mark.parametrize("kw", _FP_CASES)
def test_row_fingerprints_change_no_bit_of_the_result(lib, kw):
    """The moved-row sweeps skip a row whose new responsibilities carry the fingerprint of the old ones
    (softmax_cached_kernel, qhash_step).  With LC_SPLIT_NO_QHASH every old value is read and compared instead: the two
    runs must agree in every bit of qZ and F -- a fingerprint collision (round 4's linear sum had systematic ones: ADVICE
    r4) leaves a row with stale values and shows here."""
    a = _run_snippet({"LC_SPLIT_DELTA_FORCE": "1"}, **kw)
    b = _run_snippet({"LC_SPLIT_DELTA_FORCE": "1", "LC_SPLIT_NO_QHASH": "1"}, **kw)
    assert a["K"] == b["K"] and a["rounds"] == b["rounds"]
    assert a["Fhex"] == b["Fhex"] and a["qsha"] == b["qsha"]


@pytest.mark.parametrize("kw", [_FP_CASES[1], _FP_CASES[3], dict(seed=10, K=8, D=24, N=250000, J=1, scale=2.5)])
def test_row_wise_resync_of_a_trial_copy_changes_no_bit_of_the_result(lib, kw):
    """A split candidate works on a copy of the responsibilities (cluster.cpp:468-470).  After a rejected candidate that copy
    differs from the original in the candidate's rows only, and the next candidate's copy is made by bringing those rows back
    (Context::qz_clone_to_alt / qz_resync_kernel: rows whose fingerprints differ) instead of copying everything.  With
    LC_SPLIT_FULL_CLONE (test-hooks library) every candidate gets a full copy: same rounds, same bits -- and the row-wise path
    must really have carried most candidates."""
    env = {"LC_LIB_PATH": HOOKED, "LC_SPLIT_DELTA_FORCE": "1", "LC_TRACE_PHASES": "1"}
    a = _run_snippet(env, **kw)
    b = _run_snippet(dict(env, LC_SPLIT_FULL_CLONE="1"), **kw)
    assert a["K"] == b["K"] and a["rounds"] == b["rounds"]
    assert a["Fhex"] == b["Fhex"] and a["qsha"] == b["qsha"]
    fa, fb = a["_stderr"].count("[clone] full copy"), b["_stderr"].count("[clone] full copy")
    # (the first candidate of a round meets a copy of another width -- one full copy per round; every later one is row-wise)
    assert fa <= len(a["rounds"]) + 1 and fb > fa, (fa, fb, len(a["rounds"]))


@pytest.mark.parametrize("kw", _FP_CASES)
def test_every_writer_of_the_responsibilities_keeps_the_fingerprints_honest(lib, kw):
    """Test-hooks library, LC_TEST_VERIFY_QHASH: before a sweep trusts the stored fingerprints, every one of them is
    recomputed from the buffer (qhash_verify_kernel) and a row whose fingerprint is not its own aborts the learner.  This
    covers every writer of qZ on the paths cluster() really takes (qz_set, the E-step kernels, split_init, keep_columns,
    clone / swap): each must clear hash_ok or mark the rows it rewrites.  LC_TEST_QHASH_KEEP_STALE makes ensure_qz
    "forget" to clear the flag -- the same run must then fail, which shows the check has teeth."""
    env = {"LC_LIB_PATH": HOOKED, "LC_TEST_VERIFY_QHASH": "1", "LC_SPLIT_DELTA_FORCE": "1"}
    ok = _run_snippet(env, **kw)
    assert ok["_stderr"].count("fingerprints verified") >= 3, ok["_stderr"][-1500:]
    plain = _run_snippet({"LC_SPLIT_DELTA_FORCE": "1"}, **kw)
    assert plain["qsha"] == ok["qsha"] and plain["Fhex"] == ok["Fhex"]
    bad = _run_snippet(dict(env, LC_TEST_QHASH_KEEP_STALE="1", _EXPECT_FAILURE="1"), **kw)
    assert bad["rc"] != 0 and "fingerprint that is not theirs" in bad["_stderr"], bad["_stderr"][-1500:]


@pytest.mark.parametrize("kw", [dict(seed=21, K=9, D=20, N=9000, J=1, scale=3.0), dict(seed=22, K=12, D=33, N=6000, J=1, scale=5.0),
                                dict(seed=23, K=8, D=64, N=5000, J=1, scale=2.0), dict(seed=24, K=10, D=17, N=12000, J=1, scale=3.2),
                                dict(seed=25, K=7, D=100, N=4000, J=1, scale=4.0),
                                # beyond DP = 128 the parameter stream has the wide layout (the constant table
                                # was placed with the narrow stride there) and sigma comes from the O(D^2) norm bound
                                # (enough rows for eight clusters of 150 columns to be found: the path needs K >= 6; the oracle's
                                #  own run of this shape takes half a minute of CPU and is left to the narrower cases)
                                dict(seed=26, K=8, D=150, N=16000, J=1, scale=4.0, oracle=False)])
def test_bounded_recomputation_on_small_problems_against_all_rows_and_the_oracle(lib, kw):
    """The same with the row limit of the bounded path lowered (LC_SPLIT_BOUND_MIN_ROWS, test-hooks build), so that many
    shapes walk it: every bit of the result equal to the all-rows schedule, and rounds / K / F equal to the oracle's."""
    import lc_oracle as o

    kw = dict(kw)
    with_oracle = kw.pop("oracle", True)
    env = {"LC_LIB_PATH": HOOKED, "LC_SPLIT_BOUND_MIN_ROWS": "1000", "LC_SPLIT_DELTA_FORCE": "1", "LC_TRACE_PHASES": "1"}
    on = _run_snippet(env, **kw)
    off = _run_snippet(dict(env, LC_SPLIT_NO_BOUND="1"), **kw)
    assert "bounded recomputation:" in on["_stderr"] and "bounded recomputation:" not in off["_stderr"], \
        [ln for ln in on["_stderr"].splitlines() if "bounded" in ln][:12]
    assert on["K"] == off["K"] and on["rounds"] == off["rounds"]
    assert on["Fhex"] == off["Fhex"] and on["qsha"] == off["qsha"]
    if not with_oracle:
        assert on["K"] >= 6
        return
    rng = np.random.default_rng(kw["seed"])
    mu = rng.normal(0, kw["scale"], (kw["K"], kw["D"]))
    X = mu[rng.integers(0, kw["K"], kw["N"])] + rng.normal(size=(kw["N"], kw["D"]))
    tr = []
    Fo, _, _, clo = o.learnVDP(X, trace=tr)
    assert on["K"] == len(clo) and [k for k, _ in on["rounds"]] == [k for k, _ in tr]
    assert abs(on["F"] - Fo) <= 1e-9 * abs(Fo)


@pytest.mark.parametrize("kw,runs", [(dict(seed=11, K=8, D=24, N=250000, J=1, scale=2.5), True),   # separated enough for the bound to bite
                                     (dict(seed=12, K=7, D=40, N=220000, J=1, scale=1.0), False),  # heavy overlap: the cache is given up early
                                     (dict(seed=13, K=9, D=17, N=300000, J=1, scale=4.0), True)])
def test_bounded_recomputation_changes_no_bit_of_the_result(lib, kw, runs):
    """A split candidate's recomputed columns are evaluated only for the rows they can reach (Context::recompute_bounded:
    a lower bound on the new Mahalanobis distance from the reference cluster's cached column; rows whose new
    responsibility is certain to come out as exactly 0.0 get -inf instead of their true log q~).  With
    LC_SPLIT_NO_BOUND (test-hooks library) every recomputation runs over all rows: rounds, K, every free energy, F and
    every bit of qZ must be the same -- and the bounded path must really have run."""
    on = _run_snippet({"LC_LIB_PATH": HOOKED, "LC_TRACE_PHASES": "1"}, **kw)
    off = _run_snippet({"LC_LIB_PATH": HOOKED, "LC_SPLIT_NO_BOUND": "1", "LC_TRACE_PHASES": "1"}, **kw)
    assert ("bounded recomputation:" in on["_stderr"]) == runs and "bounded recomputation:" not in off["_stderr"]
    assert on["K"] == off["K"] and on["rounds"] == off["rounds"]
    assert on["Fhex"] == off["Fhex"] and on["qsha"] == off["qsha"]
