"""The fused pass's tile deal between the two blocks of a CU (FusedLaunch::yshare): per-launch kernel time and
iteration time of a 200-iteration VBEM for several shares of the second block (0 = equal shares), each in its own process
of the test-hooks library (LC_FUSED_YSHARE is one of its switches).
Usage: python tools/fused_share_probe.py            (one line per (N, D, K, share))"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CASES = [(1_000_000, 16, 8), (1_000_000, 8, 8), (1_000_000, 2, 4), (3_000_000, 12, 5), (2_000_000, 16, 3), (700_000, 16, 8)]
SHARES = ["0", "350", "380", "410", "440"]


def one(N, D, K):
    from libcluster_amd import capi
    import bench
    mu, L = bench.mixture(D, K, 77)
    with capi.Context(0) as ctx:
        ctx.synth_groups([N], D, K, mu, L, 77)
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=5, nthreads=8)
        m.close()
        ctx.timing_enable(True)
        ctx.timing_reset()
        t0 = time.perf_counter()
        F2, tr2, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=200, nthreads=8)
        wall = time.perf_counter() - t0
        t = ctx.timing_get()
        m.close()
    print("N %8d D %2d K %2d share %4s  kernel %.4f ms  iteration %.4f ms  F %r" % (
        N, D, K, os.environ.get("LC_FUSED_YSHARE", "-"), t["fused_ms"] / max(1, t["fused_calls"]), wall / 200 * 1e3, float(tr2[-1])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 4:
        one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        hooked = str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
        for rep in range(2):
            for N, D, K in CASES:
                for sh in SHARES:
                    subprocess.run([sys.executable, __file__, str(N), str(D), str(K)],
                                   env=dict(os.environ, LC_LIB_PATH=hooked, LC_FUSED_YSHARE=sh), check=False)
