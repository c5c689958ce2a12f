"""ctypes loader for the C port of the reference arithmetic (oracle/lc_oracle_c.c).

TEST INFRASTRUCTURE ONLY (see the header of lc_oracle_c.c): used by tests and
by bench.py's cpu_baseline leg, never by the product path.  PARITY UNPINNED."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
from scipy.special import digamma

HERE = Path(__file__).resolve().parent
SRC = HERE / "lc_oracle_c.c"
LIB = HERE / "_build" / "liblc_oracle_c.so"
_dp = C.POINTER(C.c_double)
_lib = None


def build(force: bool = False) -> Path:
    LIB.parent.mkdir(exist_ok=True)
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", str(SRC), "-o", str(LIB),
                        "-lm"], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB))
        L.lco_estep.argtypes = [_dp, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.lco_suffstat.argtypes = [_dp, _dp, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def physical_cores() -> int:
    """Physical core count (SMT siblings excluded) of this host."""
    try:
        seen = set()
        for d in Path("/sys/devices/system/cpu").glob("cpu[0-9]*"):
            t = d / "topology" / "thread_siblings_list"
            if t.exists():
                seen.add(t.read_text().strip())
        if seen:
            return min(len(seen), len(os.sched_getaffinity(0)))
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def estep(X, nu, beta, m, iW, logdW, Elogpi, nthreads=1):
    """vbexpectation for one group -> (qZ [N,K], Fz)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, D = X.shape
    nu, beta, m, iW, logdW, Elogpi = (np.ascontiguousarray(a, dtype=np.float64) for a in (nu, beta, m, iW, logdW, Elogpi))
    K = nu.size
    if D > 128:
        raise ValueError("D > 128")
    sumpsi = np.array([digamma((nu[k] + 1 - np.arange(1, D + 1)) / 2).sum() for k in range(K)])
    q = np.empty((N, K))
    Fz = C.c_double()
    rc = lib().lco_estep(_p(X), N, D, K, _p(nu), _p(beta), _p(m), _p(iW), _p(logdW), _p(sumpsi), _p(Elogpi), _p(q),
                         C.byref(Fz), nthreads)
    if rc == 1:
        raise ValueError("Matrix A is not positive definite")
    if rc:
        raise MemoryError
    return q, Fz.value


def suffstat(X, qZ, nthreads=1):
    X = np.ascontiguousarray(X, dtype=np.float64)
    qZ = np.ascontiguousarray(qZ, dtype=np.float64)
    N, D = X.shape
    K = qZ.shape[1]
    Nk, xs, xxs = np.empty(K), np.empty((K, D)), np.empty((K, D, D))
    rc = lib().lco_suffstat(_p(X), _p(qZ), N, D, K, _p(Nk), _p(xs), _p(xxs), nthreads)
    if rc:
        raise MemoryError
    return Nk, xs, xxs


def vbem_fixed(X, qZ0, wfactory, clusterprior, iters, nthreads=1):
    """``iters`` VBEM iterations of cluster.cpp:198-234 on ONE group with Gauss-Wishart clusters: the two data passes
    (updateSS / vbexpectation) by the C port above with rows chunked over ``nthreads``, the M-step, the weights and
    the free energy by the numpy oracle (lc_oracle.py).  Same results as lc_oracle.vbem_fixed (tests/test_oracle_c.py)
    at sizes the numpy oracle cannot run in seconds.  Returns (F trace, qZ)."""
    import lc_oracle as o

    X = np.ascontiguousarray(X, dtype=np.float64)
    D, K = X.shape[1], qZ0.shape[1]
    w = wfactory()
    clusters = [o.GaussWish(clusterprior, D) for _ in range(K)]
    q = np.ascontiguousarray(qZ0, dtype=np.float64)
    Ftrace = []
    for _ in range(iters):
        Nk, xs, xxs = suffstat(X, q, nthreads)
        for k, c in enumerate(clusters):
            c.clearobs()
            c.addstats(Nk[k], xs[k], xxs[k])
        w.update(Nk)  # Njk = qZ.colwise().sum() of the only group (cluster.cpp:62, 211)
        for c in clusters:
            c.update()
        q, Fz = estep(X, [c.nu for c in clusters], [c.beta for c in clusters], np.stack([c.m for c in clusters]),
                      np.stack([c.iW for c in clusters]), [c.logdW for c in clusters], w.Elogweight(), nthreads)
        Ftrace.append(o.fenergy([w], clusters, Fz))
    return Ftrace, q


def host_fp64_peak_gflops(cores: int):
    """Nominal fp64 peak of `cores` cores of this host: cores x max clock x flop per cycle (two 512-bit FMA pipes = 32
    with AVX-512 as on Zen 4/5 and recent Xeons, 16 with AVX2).  Returns (GFLOP/s, description) -- an upper bound for
    putting the measured CPU rate in proportion, not a measurement."""
    flags, mhz = "", 0.0
    try:
        txt = Path("/proc/cpuinfo").read_text()
        for line in txt.splitlines():
            if line.startswith("flags") and not flags:
                flags = line
        for p in ("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq",):
            if Path(p).exists():
                mhz = float(Path(p).read_text()) / 1e3
        if not mhz:
            vals = [float(l.split(":")[1]) for l in txt.splitlines() if l.startswith("cpu MHz")]
            mhz = max(vals) if vals else 0.0
    except (OSError, ValueError):
        pass
    per_cycle = 32 if "avx512f" in flags else 16
    if not mhz:
        return None, "clock unknown"
    return cores * mhz * 1e-3 * per_cycle, f"{cores} cores x {mhz / 1e3:.2f} GHz x {per_cycle} flop/cycle"
