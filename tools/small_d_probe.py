"""Half-width against full-width instance of fused_small_kernel at D <= 8: per-launch kernel time and the
iteration time of a 100-iteration VBEM on N = 1M synthetic rows, each instance in its own process (LC_FUSED_FULL is read
once per process).  Usage: python tools/small_d_probe.py            (prints one line per (D, K, instance))"""
import os
import subprocess
import sys
import time

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
CASES = [(2, 4), (5, 8), (5, 16), (8, 8), (8, 16)]


def one(D, K):
    import numpy as np  # noqa: F401
    from libcluster_amd import capi
    import bench
    N = 1_000_000
    mu, L = bench.mixture(D, K, 77)
    with capi.Context(0) as ctx:
        ctx.synth_groups([N], D, K, mu, L, 77)
        F, tr, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=3, nthreads=8)
        m.close()
        ctx.timing_enable(True)
        ctx.timing_reset()
        t0 = time.perf_counter()
        F2, tr2, m = ctx.vbem(capi.W_STICKBREAK, fixed_iters=100, nthreads=8)
        wall = time.perf_counter() - t0
        t = ctx.timing_get()
        m.close()
    print("D %2d K %2d %-5s fused_ms %.4f iteration_ms %.4f F %r" % (D, K, "full" if os.environ.get("LC_FUSED_FULL") else "half",
                                                                  t["fused_ms"] / max(1, t["fused_calls"]), wall * 10, float(tr2[-1])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        one(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for D, K in CASES:
            for full in (False, True):
                env = dict(os.environ)
                env.pop("LC_FUSED_FULL", None)
                if full:
                    env["LC_FUSED_FULL"] = "1"  # (a switch of the test-hooks build: lck::test_switch)
                    env["LC_LIB_PATH"] = str(__import__("pathlib").Path(__file__).resolve().parents[1] / "libcluster_amd" / "lib" / "libcluster_hip_testhooks.so")
                subprocess.run([sys.executable, __file__, str(D), str(K)], env=env, check=False)
