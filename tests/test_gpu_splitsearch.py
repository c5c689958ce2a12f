"""The split search's shortcuts against the reference's literal schedule.

`split_gr` (cluster.cpp:366-495) runs a full `vbem(..., 1)` on all data for every candidate.  The HIP path keeps the
arithmetic and shares what is provably common between the candidates of a round: LL_k as a by-product of the last
E-step (no extra pass), the statistics of the unchanged columns (two-column passes), the distances of the unchanged
clusters (cached once per round).  Each shortcut has an environment switch that restores the literal schedule; the two
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


def _learn(args, env):
    e = dict(os.environ)
    e.update(env)
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
