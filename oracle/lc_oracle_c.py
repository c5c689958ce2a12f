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
