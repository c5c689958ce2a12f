"""Small and medium Gauss-Wishart shapes: per-launch time of the E-step and the statistics pass and
their fractions of the fp64 peak on ALGORITHMIC flops (K (D^2 + 4 D), K (D^2 + 3 D + 1) per row, D = the observation width,
not its padding).  Each case in its own process; environment switches (LC_SS_FEAT ...) reach the test-hooks library.
Usage: python tools/small_shapes_probe.py [N D K ...triples]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CASES = [(6_000_000, 23, 16), (6_000_000, 32, 16), (6_000_000, 23, 8), (6_000_000, 32, 8), (5_000_000, 48, 12), (5_000_000, 40, 12),
         (4_000_000, 64, 8), (4_000_000, 56, 8), (4_000_000, 64, 16), (4_000_000, 64, 4), (3_000_000, 96, 8), (3_000_000, 88, 16)]


def one(N, D, K):
    from libcluster_amd import capi
    import bench
    import ctypes
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
    fn = capi.lib().lc_statistics_kernel_name
    fn.restype = ctypes.c_char_p
    fn.argtypes = [ctypes.c_int, ctypes.c_int]
    ss = t["suffstat_ms"] / max(1, t["suffstat_calls"])
    es = t["estep_ms"] / max(1, t["estep_calls"])
    print("N %8d D %3d K %2d %-22s statistics %.3f ms (%.3f)  E-step %.3f ms (%.3f)  %s F %r" % (
        N, D, K, fn(D, K).decode(), ss, N * K * (D * D + 3 * D + 1) / ss / 1e9 / 78.6, es, N * K * (D * D + 4 * D) / es / 1e9 / 78.6,
        " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("LC_") and k != "LC_LIB_PATH"), float(tr2[-1])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        a = [int(v) for v in sys.argv[1:]]
        cases = list(zip(a[0::3], a[1::3], a[2::3])) if a else CASES
        hooked = str(ROOT / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
        for N, D, K in cases:
            subprocess.run([sys.executable, __file__, "one", str(N), str(D), str(K)],
                           env=dict(os.environ, LC_LIB_PATH=os.environ.get("LC_LIB_PATH", hooked)), check=False)
