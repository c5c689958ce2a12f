"""What an iteration of the fused pass (D <= 16) spends outside its kernel, and what waiting on the fold's completion flag
instead of the stream buys (LC_FUSED_SPIN, Context::estep_suffstat_fused): per-launch kernel time and iteration time of a
300-iteration VBEM on N = 1M synthetic rows, each setting in its own process.
Usage: python tools/fused_host_probe.py          (one line per (D, K, setting))"""
import os
import subprocess
import sys
import time

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
CASES = [(16, 8), (2, 4), (8, 8), (16, 16)]


def one(D, K):
    from libcluster_amd import capi
    import bench
    N = 1_000_000
    mu, L = bench.mixture(D, K, 77)
    with capi.Context(0) as ctx:
        ctx.synth_groups([N], D, K, mu, L, 77)
        F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=5, nthreads=8)
        m.close()
        out = []
        for timing in (True, False):
            ctx.timing_enable(timing)
            ctx.timing_reset()
            t0 = time.perf_counter()
            F2, tr2, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=300, nthreads=8)
            wall = time.perf_counter() - t0
            t = ctx.timing_get() if timing else None
            m.close()
            out.append((wall / 300 * 1e3, t["fused_ms"] / max(1, t["fused_calls"]) if t else float("nan")))
    print("D %2d K %2d spin=%s  kernel %.4f ms  iteration %.4f ms (with timing events)  %.4f ms (without)  F %r" % (
        D, K, os.environ.get("LC_FUSED_SPIN", "1"), out[0][1], out[0][0], out[1][0], float(tr2[-1])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        one(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for D, K in CASES:
            for spin in ("1", "0"):
                subprocess.run([sys.executable, __file__, str(D), str(K)], env=dict(os.environ, LC_FUSED_SPIN=spin), check=False)
