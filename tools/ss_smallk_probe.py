"""The feature-GEMM statistics kernel at FEW clusters (K <= 16: one or two cluster quads per launch, where the per-cluster
kernel is the default) against the per-cluster kernel: per-launch time of the statistics pass, each setting in its own
process of the test-hooks library (LC_SS_FEAT=2 forces the feature GEMM wherever an instance exists, 0 never).
Usage: python tools/ss_smallk_probe.py"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CASES = [(4_000_000, 64, 4), (4_000_000, 64, 8), (4_000_000, 64, 12), (4_000_000, 64, 16), (6_000_000, 32, 8), (6_000_000, 32, 16),
         (2_000_000, 128, 8), (2_000_000, 128, 16), (3_000_000, 96, 8), (3_000_000, 96, 16), (5_000_000, 48, 12)]


def one(N, D, K):
    from libcluster_amd import capi
    import bench
    mu, L = bench.mixture(D, K, 77)
    with capi.Context(0) as ctx:
        ctx.synth_groups([N], D, K, mu, L, 77)
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=2, nthreads=8)
        m.close()
        ctx.timing_enable(True)
        ctx.timing_reset()
        F2, tr2, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=8, nthreads=8)
        t = ctx.timing_get()
        m.close()
    ss = t["suffstat_ms"] / max(1, t["suffstat_calls"])
    DP = (D + 15) // 16 * 16
    print("N %8d D %3d K %2d LC_SS_FEAT=%s  statistics %.3f ms (%.3f of the fp64 peak)  E-step %.3f ms  F %r" % (
        N, D, K, os.environ.get("LC_SS_FEAT", "-"), ss, N * K * (DP * DP + 3 * DP + 1) / ss / 1e9 / 78.6,
        t["estep_ms"] / max(1, t["estep_calls"]), float(tr2[-1])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 4:
        one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        hooked = str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
        for N, D, K in CASES:
            for mode in ("0", "2"):
                subprocess.run([sys.executable, __file__, str(N), str(D), str(K)],
                               env=dict(os.environ, LC_LIB_PATH=hooked, LC_SS_FEAT=mode), check=False)
