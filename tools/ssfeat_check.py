#!/usr/bin/env python3
"""The feature-GEMM statistics kernel (suffstat_feat_kernel, D = 64, 17..32 clusters) against the per-cluster kernel
(LC_SS_FEAT=0) and numpy: N_k, s_k, S_k on ragged row counts and every cluster count of its range; then timing at the
north-star shape.  Usage: tools/ssfeat_check.py [child]"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CASES = [(1000, 64, 17), (4099, 64, 20), (777, 50, 24), (30001, 64, 29), (65536, 64, 32), (123457, 61, 32), (31, 64, 18),
         (5000, 32, 17), (9001, 20, 33), (7000, 48, 32), (6007, 40, 70), (5003, 80, 24), (4001, 70, 64), (4999, 96, 32),
         (3001, 112, 40), (6000, 128, 32), (5001, 128, 64), (2000, 100, 65), (3000, 64, 64), (2500, 64, 45),
         # D = 128: ranges of 64 clusters in one pass (16 quads), full and ragged, one and two ranges
         (3000, 128, 57), (2777, 120, 60), (2100, 128, 128), (1900, 128, 121),
         # active widths below the padded one (round 6): 28, 36, 44 columns (20 is the (9001, 20, 33) case above), 56, 88
         (5000, 27, 20), (4000, 35, 24), (4000, 44, 28), (3000, 55, 32), (2500, 85, 20)]
# few clusters (K <= 16) at D <= 64: suffstat_quad_kernel's instances (1, 2, 4 parts per quad; both active widths per layout)
QUAD_CASES = [(1000, 23, 16), (4099, 32, 8), (777, 20, 4), (30001, 24, 12), (5000, 48, 12), (123, 40, 3), (9001, 33, 16),
              (7000, 48, 16), (6007, 64, 8), (5003, 56, 4), (4001, 64, 16), (3000, 64, 12), (2500, 17, 1), (40001, 30, 5),
              (2000, 64, 2), (3001, 47, 9), (65536, 28, 13), (1027, 50, 6), (31, 64, 7), (8000, 57, 15), (3000, 43, 8), (2222, 18, 10),
              (1500, 36, 16)]
if os.environ.get("LC_SSCHECK_CASES") == "quad":
    CASES = QUAD_CASES
TIMING = [(2000000, 64, 17), (2000000, 64, 20), (2000000, 64, 24), (4000000, 32, 20), (2000000, 48, 24), (4000000, 32, 32), (4000000, 32, 40), (2000000, 48, 32), (2000000, 48, 48), (2000000, 64, 32), (2000000, 64, 28),
          (2000000, 64, 33), (2000000, 64, 36), (2000000, 64, 40), (2000000, 64, 48), (2000000, 64, 56), (2000000, 64, 64),
          (1000000, 128, 32), (1000000, 128, 28), (1000000, 128, 40), (1000000, 128, 48), (1000000, 128, 64)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa: F401
    from libcluster_amd import capi

    out = []
    for N, D, K in CASES:
        rng = np.random.default_rng(N + K)
        X = rng.normal(size=(N, D)) * 2.0 + rng.normal(size=(1, D))
        q = rng.dirichlet(np.ones(K) * 0.3, N)
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            ctx.set_qz(q)
            Nk, xs, xxs, Njk = ctx.suffstat()
        rN = q.sum(0)
        rs = q.T @ X
        rS = np.stack([(X * q[:, k, None]).T @ X for k in range(K)])  # (BLAS: the three-operand einsum took two minutes of the suite)
        out.append(dict(case=[N, D, K], sym=bool(np.array_equal(xxs, np.transpose(xxs, (0, 2, 1)))),
                        eN=float(np.max(np.abs(Nk - rN) / rN)), es=float(np.max(np.abs(xs - rs)) / np.max(np.abs(rs))),
                        eS=float(np.max(np.abs(xxs - rS)) / np.max(np.abs(rS))),
                        h=[float(Nk.sum()), float(xs.sum()), float(xxs.sum())]))
    times = []
    for N, D, K in ([] if os.environ.get("LC_SSFEAT_NOTIME") else TIMING):
        rng = np.random.default_rng(1)
        X = rng.normal(size=(N, D))
        with capi.Context(0) as ctx:
            ctx.set_data(X)
            del X
            q = rng.dirichlet(np.ones(K) * 0.3, N)
            ctx.set_qz(q)
            del q
            ctx.timing_enable(True)
            for rep in range(3):
                ctx.timing_reset()
                ctx.suffstat()
                t = ctx.timing_get()
            DP = max(16, (D + 15) // 16 * 16)
            ms = t["suffstat_ms"] / t["suffstat_calls"]
            times.append([N, D, K, ms, N * K * (DP * DP + 3 * DP + 1) / ms / 1e9 / 78.6])
    print("RESULT " + json.dumps([out, times]))
    sys.exit(0)

res = {}
tim = {}
for name, env in (("feat", {"LC_SS_FEAT": "2"}), ("percluster", {"LC_SS_FEAT": "0"})):
    # (LC_SS_FEAT is a switch of the test-hooks build of the library: lck::test_switch)
    e = dict(os.environ, LC_LIB_PATH=str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so"), **env)
    p = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=e, timeout=900)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    if p.returncode or not line:
        print(name, "FAILED", p.stderr[-2000:])
        sys.exit(1)
    res[name], tim[name] = json.loads(line[-1][7:])
ok = True
for a, b in zip(res["feat"], res["percluster"]):
    good = a["sym"] and a["eN"] < 1e-12 and a["es"] < 1e-12 and a["eS"] < 1e-12
    ok = ok and good
    print(a["case"], "feat: sym", a["sym"], f"eN {a['eN']:.1e} es {a['es']:.1e} eS {a['eS']:.1e}", "| per-cluster:",
          f"eN {b['eN']:.1e} es {b['es']:.1e} eS {b['eS']:.1e}", "OK" if good else "BAD")
print("ALL OK" if ok else "FAILURES")
for a, b in zip(tim["feat"], tim["percluster"]):
    print(f"N={a[0]} D={a[1]} K={a[2]}: feature GEMM {a[3]:8.3f} ms ({a[4]:.3f} of the fp64 peak)   per-cluster {b[3]:8.3f} ms ({b[4]:.3f})   {'FEAT' if a[3] < b[3] else 'per-cluster'} wins")
sys.exit(0 if ok else 1)
