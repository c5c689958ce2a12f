#!/usr/bin/env python3
"""Upload of a pageable 10M x 64 host matrix: the library's packer (two page-locked 32 MB buffers, two host cores) against
pinning the caller's buffer in place (hipHostRegister) and one DMA.  Usage (GPU box): tools/pin_probe.py [rows]"""
import ctypes as C
import sys
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
X = np.random.default_rng(1).normal(size=(N, 64))
nbytes = X.nbytes
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(nbytes)) == 0
p = C.c_void_p(X.ctypes.data)
for rep in range(3):
    t0 = time.perf_counter()
    r = hip.hipHostRegister(p, C.c_size_t(nbytes), 0)
    t1 = time.perf_counter()
    assert r == 0, r
    assert hip.hipMemcpy(d, p, C.c_size_t(nbytes), 1) == 0
    assert hip.hipDeviceSynchronize() == 0
    t2 = time.perf_counter()
    assert hip.hipHostUnregister(p) == 0
    t3 = time.perf_counter()
    print(f"in place: register {t1 - t0:.3f} s, copy {t2 - t1:.3f} s ({nbytes / (t2 - t1) / 1e9:.1f} GB/s), unregister {t3 - t2:.3f} s, "
          f"total {t3 - t0:.3f} s", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(d, p, C.c_size_t(nbytes), 1) == 0
    assert hip.hipDeviceSynchronize() == 0
    t1 = time.perf_counter()
    print(f"hipMemcpy from pageable memory: {t1 - t0:.3f} s ({nbytes / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from libcluster_amd import capi  # noqa: E402

with capi.Context(0) as ctx:
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.set_data(X)
        t1 = time.perf_counter()
        print(f"lc_ctx_set_data (packer): {t1 - t0:.3f} s ({nbytes / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
