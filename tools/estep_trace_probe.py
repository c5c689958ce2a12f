#!/usr/bin/env python3
"""Where does the E-step's wall time go across the chip?  A library built with -DLC_ES_TRACE (tools/variants.py build
trace -DLC_ES_TRACE) records, per block: XCD, hardware slot, wall-clock start / end (100 MHz) and shader clocks.
Prints per-XCD: blocks, first start, last end, busy span, mean block duration, mean shader clock."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("LC_LIB_PATH", str(ROOT / "tools" / "variants" / "trace.so"))
os.environ["LC_ALLOW_STALE_LIB"] = "1"
import torch  # noqa: F401,E402
import bench  # noqa: E402
from libcluster_amd import capi  # noqa: E402

N, D, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 64, 32)
mu, L = bench.mixture(D, K, 1004)
with capi.Context(0) as ctx:
    ctx.synth(N, D, K, mu, L, 1004, 0, 0.9)
    F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=4, nthreads=8)
    m.close()
    ctx.synchronize()
    nb = min(65536, -(-N // 16 // 12))
    buf = np.zeros((nb, 5), dtype=np.int64)
    fn = capi.lib().lc_debug_estep_trace
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf.ctypes.data, nb) == 0
xcc, hw, w0, w1, clk = buf.T
t0 = w0.min()
print(f"N={N} D={D} K={K}: {nb} blocks; kernel span {(w1.max() - t0) / 100:.1f} us (100 MHz wall clock)")
print("xcd  blocks  first_start_us  last_end_us  mean_block_us  mean_clock_GHz   p5/p95 block_us")
for x in sorted(set(xcc.tolist())):
    s = xcc == x
    dur = (w1[s] - w0[s]) / 100.0
    ghz = clk[s] / (w1[s] - w0[s]) * 0.1
    print(f"{x:3d} {s.sum():7d} {(w0[s].min() - t0) / 100:14.1f} {(w1[s].max() - t0) / 100:12.1f} {dur.mean():14.2f} {ghz.mean():15.3f}   {np.percentile(dur, 5):.1f}/{np.percentile(dur, 95):.1f}")
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("per (XCD, SE, CU): blocks served  min..max:", end=" ")
keys = xcc * 1000 + se * 16 + cu
u, c = np.unique(keys, return_counts=True)
print(len(u), "CUs;", c.min(), "..", c.max())
# the last blocks to finish
order = np.argsort(w1)[-12:]
print("last blocks to end (block id, xcd, end_us, dur_us):", [(int(i), int(xcc[i]), round((w1[i] - t0) / 100, 1), round((w1[i] - w0[i]) / 100, 1)) for i in order])
