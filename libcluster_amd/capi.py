"""ctypes binding of include/libcluster_hip.h (plumbing only: every function
here forwards to the C-ABI; there is no Python or CPU implementation of the
data path).  Loading fails loudly if the shared library has not been built."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libcluster_hip.so"
if __import__("os").environ.get("LC_LIB_PATH"):  # kernel experiments (tools/variants.py): another build of the library
    LIB_PATH = Path(__import__("os").environ["LC_LIB_PATH"])
HEADER = PKG.parent / "include" / "libcluster_hip.h"

LC_OK, LC_EINVAL, LC_ERUNTIME, LC_EHIP, LC_EDOMAIN = range(5)
W_DIRICHLET, W_STICKBREAK, W_GDIRICHLET = 0, 1, 2
ALGO_VDP, ALGO_BGMM, ALGO_GMC, ALGO_SGMC, ALGO_DGMM, ALGO_BEMM, ALGO_DGMC, ALGO_EGMC = range(8)
C_GAUSSWISH, C_NORMGAMMA, C_EXPGAMMA = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId: rank 0 calls it and ships the bytes to the other ranks."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().lc_comm_unique_id(buf))
    return buf.raw


def rccl_available() -> bool:
    return bool(lib().lc_comm_rccl_available())


class HipError(RuntimeError):
    """HIP runtime failure (including: no GPU -- there is no CPU fallback)."""


class DomainError(ArithmeticError):
    pass


_lib = None


def declared_symbols() -> list[str]:
    """Every function the public header declares (used by the symbol test)."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(n for n in names if n != "lc_allreduce_fn"))


def _load_checked() -> C.CDLL:
    """dlopen the library and compare the source hash compiled into it (lc_source_hash) with the tree it sits in: a
    stale binary (sources edited after the last build; the .so is git-ignored and travels prebuilt) is rebuilt when
    hipcc is there and refused otherwise.  LC_ALLOW_STALE_LIB=1 skips the check, and so does LC_LIB_PATH (a user-supplied
    library is loaded as it is)."""
    import os

    from . import build as _build

    def load():
        L = C.CDLL(str(LIB_PATH))
        try:
            L.lc_source_hash.restype = C.c_char_p
            return L, L.lc_source_hash().decode()
        except AttributeError:  # a binary from before the guard existed
            return L, "missing"

    L, have = load()
    # LC_LIB_PATH names another build of the library (kernel experiments): it is the caller's, never compared with this
    # tree and never rebuilt into or over
    if os.environ.get("LC_ALLOW_STALE_LIB") or os.environ.get("LC_LIB_PATH"):
        return L
    want = _build.source_hash()
    if have == want:
        return L
    # dlopen caches by path: a rebuilt file must be loaded under a new name, so rebuild BEFORE anything else binds to
    # the stale handle, then load a private copy of the fresh file
    try:
        _build.build()
    except Exception as e:  # noqa: BLE001
        raise ImportError(f"{LIB_PATH} is stale (built from sources {have}, tree is {want}) and rebuilding failed: {e}. "
                          "Run `python -m libcluster_amd.build`.") from e
    import shutil
    import tempfile

    tmp = Path(tempfile.mkdtemp(prefix="lc_fresh_")) / LIB_PATH.name
    shutil.copy2(LIB_PATH, tmp)
    L2 = C.CDLL(str(tmp))
    L2.lc_source_hash.restype = C.c_char_p
    if L2.lc_source_hash().decode() != want:
        raise ImportError(f"{LIB_PATH} is stale and the rebuild did not pick up the current sources")
    import sys

    print(f"libcluster_amd: rebuilt a stale {LIB_PATH.name} (was {have}, now {want})", file=sys.stderr)
    return L2


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m libcluster_amd.build` "
            "(hipcc, --offload-arch=gfx950).  There is no Python/CPU fallback for the E-step."
        )
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64, and
    # a process that initialises both that copy and the system ROCm copy loses the GPU in the second
    # one ("no ROCm-capable device").  Loading torch first makes the C-ABI library bind to the runtime
    # torch uses (same SONAME), which is also what lets torch streams and RCCL see our buffers.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = _load_checked()
    L.lc_last_error.restype = C.c_char_p
    L.lc_comm_unique_id.argtypes = [C.c_void_p]
    L.lc_ctx_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.lc_ctx_comm_init_host.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.lc_ctx_comm_free.argtypes = [C.c_void_p]
    L.lc_ctx_comm_info.argtypes = [C.c_void_p, c_int_p, c_int_p, C.POINTER(C.c_char_p)]
    L.lc_ctx_allreduce.argtypes = [C.c_void_p, c_double_p, C.c_int]
    for f in ("lc_const_converge", "lc_const_fengydel", "lc_const_zerocutoff"):
        getattr(L, f).restype = C.c_double
    L.lc_digamma.restype = C.c_double
    L.lc_digamma.argtypes = [C.c_double]
    L.lc_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.lc_ctx_destroy.argtypes = [C.c_void_p]
    L.lc_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.lc_ctx_synchronize.argtypes = [C.c_void_p]
    L.lc_ctx_dims.argtypes = [C.c_void_p, c_int_p, c_int_p, c_int64_p, c_int_p]
    L.lc_ctx_set_data.argtypes = [C.c_void_p, C.c_int, C.POINTER(c_double_p), c_int64_p, C.c_int, C.c_int64, C.c_int64]
    L.lc_ctx_synth.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, c_double_p, c_double_p, C.c_uint64,
                               C.c_int64, C.c_double]
    L.lc_ctx_synth_groups.argtypes = [C.c_void_p, C.c_int, c_int64_p, C.c_int, C.c_int, c_double_p, c_double_p,
                                      c_double_p, C.c_uint64, c_int64_p, C.c_double]
    L.lc_ctx_set_sharding.argtypes = [C.c_void_p, C.c_int]
    L.lc_ctx_set_skip_zero.argtypes = [C.c_void_p, C.c_int]
    L.lc_ctx_get_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, c_double_p]
    L.lc_ctx_set_qz.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int, C.c_int64, C.c_int64]
    L.lc_ctx_get_qz.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int64, C.c_int64]
    L.lc_ctx_get_qz_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, c_double_p, C.c_int64, C.c_int64]
    L.lc_ctx_fill_qz.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.lc_estep.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_estep_posterior.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                     c_double_p, c_ubyte_p, c_double_p, c_double_p]
    L.lc_eloglike.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_mahaldist.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
    L.lc_suffstat.argtypes = [C.c_void_p, c_ubyte_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_suffstat_diag.argtypes = [C.c_void_p, c_ubyte_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_estep_diag.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int,
                                c_double_p, c_double_p]
    L.lc_colsums.argtypes = [C.c_void_p, c_double_p]
    L.lc_ctx_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    L.lc_ctx_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.lc_ctx_timing_reset.argtypes = [C.c_void_p]
    L.lc_ctx_timing_get.argtypes = [C.c_void_p, c_double_p, c_int64_p, c_double_p, c_int64_p]
    L.lc_vbem.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_uint, c_double_p, c_int_p, c_double_p, C.c_int]
    L.lc_learn.argtypes = [C.c_int, C.c_int, C.POINTER(c_double_p), c_int64_p, C.c_int, C.c_int64, C.c_int64,
                           C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int,
                           C.POINTER(C.c_void_p), c_double_p]
    L.lc_learn_w.argtypes = [C.c_int, C.c_int, C.POINTER(c_double_p), c_int64_p, C.c_int, C.c_int64, C.c_int64,
                             C.c_double, c_double_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int,
                             C.POINTER(C.c_void_p), c_double_p]
    L.lc_cluster.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                             C.c_uint, C.POINTER(C.c_void_p), c_double_p]
    L.lc_prune.argtypes = [C.c_void_p, C.c_void_p, C.c_int, c_int_p]
    L.lc_model_free.argtypes = [C.c_void_p]
    L.lc_model_dims.argtypes = [C.c_void_p, c_int_p, c_int_p, c_int_p]
    L.lc_model_rounds.argtypes = [C.c_void_p, c_int_p]
    L.lc_model_round.argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_double_p, C.c_int]
    L.lc_model_get_qz.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int64, C.c_int64]
    L.lc_model_weights.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p]
    L.lc_model_cluster.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                   c_double_p, c_double_p]
    L.lc_model_kinds.argtypes = [C.c_void_p, c_int_p, c_int_p]
    L.lc_model_fenergy.argtypes = [C.c_void_p, c_double_p, c_double_p]
    L.lc_weights_update.argtypes = [C.c_int, C.c_double, c_double_p, C.c_int, c_double_p, c_double_p]
    L.lc_gw_mstep.argtypes = [C.c_double, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                              c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_learn_topic.argtypes = [C.c_int, c_int_p, C.POINTER(c_double_p), c_int64_p, C.c_int, C.c_int64, C.c_int64,
                                 C.POINTER(c_double_p), C.c_int, C.POINTER(c_double_p), C.c_double, C.c_double,
                                 C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(C.c_void_p), c_double_p]
    L.lc_learn_topic_dist.argtypes = [C.c_int, c_int_p, C.POINTER(c_double_p), c_int64_p, C.c_int, C.c_int64, C.c_int64,
                                      C.POINTER(c_double_p), C.c_int, C.POINTER(c_double_p), C.c_double, C.c_double,
                                      C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p, ALLREDUCE_FN,
                                      C.c_void_p, C.POINTER(C.c_void_p), c_double_p]
    L.lc_tmodel_free.argtypes = [C.c_void_p]
    L.lc_tmodel_dims.argtypes = [C.c_void_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]
    L.lc_tmodel_get_qy.argtypes = [C.c_void_p, C.c_int, c_double_p]
    L.lc_tmodel_get_qz.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int64, C.c_int64]
    L.lc_tmodel_get_qz_all.argtypes = [C.c_void_p, c_double_p]
    L.lc_model_get_qz_all.argtypes = [C.c_void_p, c_double_p]
    L.lc_ctx_get_qz_all.argtypes = [C.c_void_p, c_double_p]
    L.lc_tmodel_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p]
    L.lc_tmodel_cluster.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                    c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_tmodel_rounds.argtypes = [C.c_void_p, c_int_p]
    L.lc_tmodel_round.argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p, c_double_p, C.c_int]
    L.lc_ng_mstep.argtypes = [C.c_double, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                              c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.lc_eg_mstep.argtypes = [C.c_double, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                              c_double_p, c_double_p]
    _lib = L
    return L


def check(rc: int) -> None:
    """Map lc_status to the exception classes the reference throws."""
    if rc == LC_OK:
        return
    msg = lib().lc_last_error().decode("utf-8", "replace")
    if rc == LC_EINVAL:
        raise ValueError(msg)  # std::invalid_argument
    if rc == LC_EHIP:
        raise HipError(msg)
    if rc == LC_EDOMAIN:
        raise DomainError(msg)  # std::domain_error
    raise RuntimeError(msg)  # std::runtime_error


def dptr(a):
    if a is None:
        return None
    assert a.dtype == np.float64
    return a.ctypes.data_as(c_double_p)


def _strides(a: np.ndarray):
    assert a.ndim == 2 and a.dtype == np.float64
    return a.strides[0] // 8, a.strides[1] // 8


class Context:
    """lc_ctx: device-resident observations + qZ (one per data set / rank)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        self._cb = None
        check(lib().lc_ctx_create(device, C.c_void_p(stream or 0), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().lc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- data ---------------------------------------------------------------
    def set_data(self, X):
        """X: (N, D) array or list of (N_j, D) arrays, any strides (no copy)."""
        Xs = [X] if isinstance(X, np.ndarray) else list(X)
        Xs = [np.asarray(x, dtype=np.float64) for x in Xs]
        D = Xs[0].shape[1]
        for x in Xs:
            if x.ndim != 2 or x.shape[1] != D:
                raise ValueError("X dimensions are inconsistent between groups!")
        st = {_strides(x) for x in Xs}
        if len(st) == 1 and all(x.shape[0] > 1 for x in Xs) and min(next(iter(st))) > 0:
            rs, cs = next(iter(st))  # e.g. all column-major (Eigen default) or all row-major: no copy
        else:
            Xs = [np.ascontiguousarray(x) for x in Xs]
            rs, cs = D, 1
        J = len(Xs)
        ptrs = (c_double_p * J)(*[dptr(x) for x in Xs])
        Nj = (C.c_int64 * J)(*[x.shape[0] for x in Xs])
        check(lib().lc_ctx_set_data(self._h, J, ptrs, Nj, D, rs, cs))
        self._X = Xs  # keep alive

    def synth(self, N, D, K, mu, L, seed, row_offset=0, hard=0.9):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        L = np.ascontiguousarray(L, dtype=np.float64)
        assert mu.shape == (K, D) and L.shape == (K, D, D)
        check(lib().lc_ctx_synth(self._h, N, D, K, dptr(mu), dptr(L), seed, row_offset, hard))

    def synth_groups(self, Nj, D, K, mu, L, seed, mix=None, group_ids=None, hard=0.9):
        """J groups of Nj rows; mix: (J, K) mixing proportions per group (None = uniform)."""
        Nj = np.ascontiguousarray(Nj, dtype=np.int64)
        J = Nj.size
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        L = np.ascontiguousarray(L, dtype=np.float64)
        cdf = None
        if mix is not None:
            cdf = np.ascontiguousarray(np.cumsum(np.asarray(mix, dtype=np.float64), axis=1))
            assert cdf.shape == (J, K)
        gid = None if group_ids is None else np.ascontiguousarray(group_ids, dtype=np.int64)
        check(lib().lc_ctx_synth_groups(self._h, J, Nj.ctypes.data_as(c_int64_p), D, K, dptr(mu), dptr(L), dptr(cdf),
                                        seed, None if gid is None else gid.ctypes.data_as(c_int64_p), hard))

    def set_skip_zero(self, on: bool = True):
        """Exact: leave out (4-row step, cluster) pairs with all-zero responsibilities in the statistics pass."""
        check(lib().lc_ctx_set_skip_zero(self._h, int(on)))

    def set_sharding(self, whole_groups: bool):
        check(lib().lc_ctx_set_sharding(self._h, int(whole_groups)))

    def dims(self):
        J, D, K, N = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        check(lib().lc_ctx_dims(self._h, C.byref(J), C.byref(D), C.byref(N), C.byref(K)))
        return J.value, D.value, N.value, K.value

    def get_rows(self, j, row0, n):
        _, D, _, _ = self.dims()
        out = np.empty((n, D))
        check(lib().lc_ctx_get_rows(self._h, j, row0, n, dptr(out)))
        return out

    def set_qz(self, qZ):
        qs = [qZ] if isinstance(qZ, np.ndarray) else list(qZ)
        K = qs[0].shape[1]
        for j, q in enumerate(qs):
            q = np.asarray(q, dtype=np.float64)
            if q.shape[1] != K:
                raise ValueError("qZ groups must have the same number of columns")
            if q.shape[0] == 0:
                q = np.zeros((1, K))[:0]
                check(lib().lc_ctx_set_qz(self._h, j, dptr(np.zeros(1)), K, K, 1))
                continue
            rs, cs = _strides(q)
            check(lib().lc_ctx_set_qz(self._h, j, dptr(q), K, rs, cs))

    def get_qz(self, rows_per_group):
        """qZ of every group (one device-to-host transfer) -> list of (N_j, K) arrays."""
        _, _, _, K = self.dims()
        rows = [int(n) for n in rows_per_group]
        allq = np.zeros((sum(rows), K))
        if allq.size:
            check(lib().lc_ctx_get_qz_all(self._h, dptr(allq)))
        return [allq[o - n:o] for n, o in zip(rows, np.cumsum(rows))]

    def get_qz_colmajor(self, rows_per_group):
        """The same through the column-major bulk getter (what Eigen callers use): list of Fortran-ordered (N_j, K)."""
        _, _, _, K = self.dims()
        out = [np.zeros((int(n), K), order="F") for n in rows_per_group]
        ptrs = (c_double_p * len(out))(*[dptr(a) if a.size else None for a in out])
        if sum(a.size for a in out):
            fn = lib().lc_ctx_get_qz_all_colmajor
            fn.argtypes = [C.c_void_p, C.POINTER(c_double_p)]
            check(fn(self._h, ptrs))
        return out

    def get_qz_rows(self, j, row0, n):
        _, _, _, K = self.dims()
        q = np.empty((n, K))
        if n:
            check(lib().lc_ctx_get_qz_rows(self._h, j, row0, n, dptr(q), K, 1))
        return q

    def fill_qz(self, K, value=1.0):
        check(lib().lc_ctx_fill_qz(self._h, K, value))

    # -- hot path -------------------------------------------------------------
    def estep_posterior(self, nu, beta, m, iW, logdW, Elogpi, active=None, want_ll=True):
        K = len(nu)
        nu, beta, m, iW, logdW, Elogpi = (np.ascontiguousarray(a, dtype=np.float64)
                                          for a in (nu, beta, m, iW, logdW, Elogpi))
        Fz = C.c_double()
        ll = np.zeros(K) if want_ll else None
        act = None
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.uint8)
            act = active.ctypes.data_as(c_ubyte_p)
        check(lib().lc_estep_posterior(self._h, K, dptr(nu), dptr(beta), dptr(m), dptr(iW), dptr(logdW),
                                       dptr(Elogpi), act, C.byref(Fz), dptr(ll)))
        return Fz.value, ll

    def eloglike(self, nu, beta, m, iW, logdW, rows_per_group):
        """K x GaussWish::Eloglike for every group -> list of (N_j, K) arrays."""
        K = len(nu)
        nu, beta, m, iW, logdW = (np.ascontiguousarray(a, dtype=np.float64) for a in (nu, beta, m, iW, logdW))
        check(lib().lc_eloglike(self._h, K, dptr(nu), dptr(beta), dptr(m), dptr(iW), dptr(logdW)))
        return self.get_qz(rows_per_group)

    def estep(self, A, m, c):
        A, m, c = (np.ascontiguousarray(a, dtype=np.float64) for a in (A, m, c))
        K = A.shape[0]
        Fz = C.c_double()
        ll = np.zeros(K)
        check(lib().lc_estep(self._h, K, dptr(A), dptr(m), dptr(c), C.byref(Fz), dptr(ll)))
        return Fz.value, ll

    def mahaldist(self, mu, A):
        """probutils::mahaldist for every resident row -> (N_total,) array."""
        J, D, N, _ = self.dims()
        mu, A = (np.ascontiguousarray(v, dtype=np.float64) for v in (mu, A))
        out = np.zeros(N)
        check(lib().lc_mahaldist(self._h, dptr(mu), dptr(A), dptr(out)))
        return out

    def suffstat(self, smask=None):
        J, D, _, K = self.dims()
        Nk, xs, xxs, Njk = np.zeros(K), np.zeros((K, D)), np.zeros((K, D, D)), np.zeros((J, K))
        sm = None
        if smask is not None:
            smask = np.ascontiguousarray(smask, dtype=np.uint8)
            sm = smask.ctypes.data_as(c_ubyte_p)
        check(lib().lc_suffstat(self._h, sm, dptr(Nk), dptr(xs), dptr(xxs), dptr(Njk)))
        return Nk, xs, xxs, Njk

    def suffstat_diag(self, smask=None, second=True):
        """NormGamma / ExpGamma statistics: N_k, sum q x, sum q x.^2 (None when second=False), N_jk."""
        J, D, _, K = self.dims()
        Nk, xs, Njk = np.zeros(K), np.zeros((K, D)), np.zeros((J, K))
        xxs = np.zeros((K, D)) if second else None
        sm = None
        if smask is not None:
            smask = np.ascontiguousarray(smask, dtype=np.uint8)
            sm = smask.ctypes.data_as(c_ubyte_p)
        check(lib().lc_suffstat_diag(self._h, sm, dptr(Nk), dptr(xs), dptr(xxs), dptr(Njk)))
        return Nk, xs, xxs, Njk

    def estep_diag(self, a, w2, w1, c, raw=False):
        a, w2, w1, c = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, w2, w1, c))
        K = a.shape[0]
        Fz = C.c_double()
        ll = np.zeros(K)
        check(lib().lc_estep_diag(self._h, K, dptr(a), dptr(w2), dptr(w1), dptr(c), int(raw), C.byref(Fz), dptr(ll)))
        return Fz.value, ll

    def colsums(self):
        J, _, _, K = self.dims()
        out = np.zeros((J, K))
        check(lib().lc_colsums(self._h, dptr(out)))
        return out

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> None; sums in place across ranks."""
        if fn is None:
            self._cb = None
            check(lib().lc_ctx_set_allreduce(self._h, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(user, buf, count, stream):
            try:
                fn(buf or 0, count, stream or 0)
                return 0
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(tramp)
        check(lib().lc_ctx_set_allreduce(self._h, self._cb, None))

    # -- native collectives (lc_comm.cpp) ---------------------------------------
    def comm_init_rccl(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on this context's device with the id rank 0 got from comm_unique_id()."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(unique_id, COMM_ID_BYTES)
        check(lib().lc_ctx_comm_init_rccl(self._h, buf, rank, world))

    def comm_init_host(self, name: str, rank: int, world: int):
        """Host-staged communicator over the shared-memory object /lc_comm_<name> (any placement of ranks)."""
        check(lib().lc_ctx_comm_init_host(self._h, name.encode(), rank, world))

    def comm_free(self):
        check(lib().lc_ctx_comm_free(self._h))

    def comm_info(self):
        r, w, k = C.c_int(), C.c_int(), C.c_char_p()
        check(lib().lc_ctx_comm_info(self._h, C.byref(r), C.byref(w), C.byref(k)))
        return {"rank": r.value, "world": w.value, "kind": (k.value or b"none").decode()}

    def allreduce(self, values):
        """Sum a small host vector over the ranks (through the context's communicator / hook)."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(lib().lc_ctx_allreduce(self._h, dptr(v), v.size))
        return v

    def timing_enable(self, on=True):
        check(lib().lc_ctx_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(lib().lc_ctx_timing_reset(self._h))

    def timing_get(self):
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_int64(), C.c_int64()
        check(lib().lc_ctx_timing_get(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        f, nf = C.c_double(), C.c_int64()
        fn = lib().lc_ctx_timing_get_fused
        fn.argtypes = [C.c_void_p, c_double_p, c_int64_p]
        check(fn(self._h, C.byref(f), C.byref(nf)))
        return {"estep_ms": a.value, "estep_calls": na.value, "suffstat_ms": b.value, "suffstat_calls": nb.value,
                "fused_ms": f.value, "fused_calls": nf.value}

    TIMING_FIELDS = ("estep_ms", "estep_calls", "suffstat_ms", "suffstat_calls", "fused_ms", "fused_calls",
                     "allreduce_ms", "allreduce_calls", "host_stats_ms", "host_mstep_ms", "host_estep_ms",
                     "host_fenergy_ms", "host_iters", "estep_diag_mfma_calls")

    def timing_get_all(self):
        """Kernel, collective and host-phase times since the last reset (lc_ctx_timing_get_all)."""
        out = (C.c_double * len(self.TIMING_FIELDS))()
        fn = lib().lc_ctx_timing_get_all
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        check(fn(self._h, out, len(self.TIMING_FIELDS)))
        return {k: (int(v) if k.endswith(("_calls", "_iters")) else float(v)) for k, v in zip(self.TIMING_FIELDS, out)}

    def synchronize(self):
        check(lib().lc_ctx_synchronize(self._h))

    def vbem(self, wkind, wprior=1.0, clusterprior=1.0, maxit=-1, sparse=False, fixed_iters=-1, verbose=False,
             nthreads=1, model=None, ntrace=4096, ckind=C_GAUSSWISH):
        mh = model._h if model is not None else C.c_void_p()
        F, nit = C.c_double(), C.c_int()
        tr = np.zeros(ntrace)
        check(lib().lc_vbem(self._h, C.byref(mh), wkind, ckind, wprior, clusterprior, maxit, int(sparse), fixed_iters,
                            int(verbose), nthreads, C.byref(F), C.byref(nit), dptr(tr), ntrace))
        if model is None:
            model = Model(mh, ctx=self)
        return F.value, tr[: nit.value].copy(), model


    def prune(self, model, verbose=False):
        """prune_clusters (cluster.cpp:505-552) on `model` and this context's qZ -> number of clusters removed."""
        n = C.c_int()
        check(lib().lc_prune(self._h, model._h, int(verbose), C.byref(n)))
        return n.value

    def cluster(self, wkind, wprior=1.0, clusterprior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1,
                ckind=C_GAUSSWISH):
        """cluster() (model selection) on the data resident in this context -> (F, Model)."""
        mh, F = C.c_void_p(), C.c_double()
        check(lib().lc_cluster(self._h, wkind, ckind, wprior, clusterprior, maxclusters, int(sparse), int(verbose),
                               nthreads, C.byref(mh), C.byref(F)))
        return F.value, Model(mh, ctx=self)


class Model:
    """lc_model: weights + clusters (+ access to the context's qZ)."""

    def __init__(self, handle, ctx=None):
        self._h = handle
        self._ctx = ctx  # keep a borrowed context alive

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().lc_model_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dims(self):
        J, K, D = C.c_int(), C.c_int(), C.c_int()
        check(lib().lc_model_dims(self._h, C.byref(J), C.byref(K), C.byref(D)))
        return J.value, K.value, D.value

    def rounds(self):
        n = C.c_int()
        check(lib().lc_model_rounds(self._h, C.byref(n)))
        out = []
        for r in range(n.value):
            K, ni = C.c_int(), C.c_int()
            check(lib().lc_model_round(self._h, r, C.byref(K), C.byref(ni), None, 0))
            F = np.zeros(ni.value)
            check(lib().lc_model_round(self._h, r, None, None, dptr(F), ni.value))
            out.append((K.value, F.tolist()))
        return out

    def qz_all(self, rows):
        """All groups with one transfer -> list of (N_j, K) views."""
        _, K, _ = self.dims()
        allq = np.zeros((int(sum(rows)), K))
        if allq.size:
            check(lib().lc_model_get_qz_all(self._h, dptr(allq)))
        return [allq[o - n:o] for n, o in zip(rows, np.cumsum(rows))]

    def qz(self, j, n):
        _, K, _ = self.dims()
        q = np.empty((n, K))
        if n:
            check(lib().lc_model_get_qz(self._h, j, dptr(q), K, 1))
        return q

    def weights(self, j):
        _, K, _ = self.dims()
        e, n = np.zeros(K), np.zeros(K)
        check(lib().lc_model_weights(self._h, j, dptr(e), dptr(n)))
        return e, n

    def kinds(self):
        w, c = C.c_int(), C.c_int()
        check(lib().lc_model_kinds(self._h, C.byref(w), C.byref(c)))
        return w.value, c.value

    def cluster(self, k):
        """Posterior of cluster k.  GaussWish: mean, cov (D x D), nu, beta, iW, logdW.  NormGamma: mean, cov (D,
        = L*nu as the reference's getcov), nu, beta, L, logL.  ExpGamma: rate (= a*ib), a, ib, logb."""
        _, _, D = self.dims()
        ck = self.kinds()[1]
        N, nu, beta, logdW = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        if ck == C_GAUSSWISH:
            mean, cov, iW = np.zeros(D), np.zeros((D, D)), np.zeros((D, D))
        else:
            mean, cov, iW = np.zeros(D), (np.zeros(D) if ck == C_NORMGAMMA else None), np.zeros(D)
        check(lib().lc_model_cluster(self._h, k, C.byref(N), dptr(mean), dptr(cov), C.byref(nu), C.byref(beta),
                                     dptr(iW), C.byref(logdW)))
        if ck == C_GAUSSWISH:
            return {"N": N.value, "mean": mean, "cov": cov, "nu": nu.value, "beta": beta.value, "iW": iW,
                    "logdW": logdW.value}
        if ck == C_NORMGAMMA:
            return {"N": N.value, "mean": mean, "cov": cov, "nu": nu.value, "beta": beta.value, "L": iW,
                    "logL": logdW.value}
        return {"N": N.value, "rate": mean, "a": nu.value, "ib": iW, "logb": logdW.value}

    def fenergy(self):
        J, K, _ = self.dims()
        Fw, Fc = np.zeros(J), np.zeros(K)
        check(lib().lc_model_fenergy(self._h, dptr(Fw), dptr(Fc)))
        return Fw, Fc


def learn(algo, X, wprior=1.0, clusterprior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1, device=0,
          wprior_j=None):
    """wprior_j: one weight prior per group (the priors the caller's weight objects carry into the multi-group
    learners: learnSGMC's Dirichlet alphas); None = defaults."""
    Xs = [X] if isinstance(X, np.ndarray) else list(X)
    Xs = [np.ascontiguousarray(x, dtype=np.float64) for x in Xs]
    J, D = len(Xs), Xs[0].shape[1]
    ptrs = (c_double_p * J)(*[dptr(x) for x in Xs])
    Nj = (C.c_int64 * J)(*[x.shape[0] for x in Xs])
    mh, F = C.c_void_p(), C.c_double()
    wj = None
    if wprior_j is not None:
        wj = np.ascontiguousarray(wprior_j, dtype=np.float64)
        if wj.shape != (J,):
            raise ValueError("wprior_j needs one value per group")
    check(lib().lc_learn_w(algo, J, ptrs, Nj, D, D, 1, wprior, dptr(wj), clusterprior, maxclusters, int(sparse),
                           int(verbose), nthreads, device, C.byref(mh), C.byref(F)))
    return F.value, Model(mh), [x.shape[0] for x in Xs]


def trim_cache():
    """Return every cached device / page-locked block to the driver."""
    check(lib().lc_trim_cache())


def weights_update(wkind, Nk, wprior=1.0):
    Nk = np.ascontiguousarray(Nk, dtype=np.float64)
    e, f = np.zeros(Nk.size), C.c_double()
    check(lib().lc_weights_update(wkind, wprior, dptr(Nk), Nk.size, dptr(e), C.byref(f)))
    return e, f.value


def gw_mstep(clustwidth, Ns, xs, xxs):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    xxs = np.ascontiguousarray(xxs, dtype=np.float64)
    D = xs.size
    nu, beta, logdW, fe, cst = (C.c_double() for _ in range(5))
    m, iW, A = np.zeros(D), np.zeros((D, D)), np.zeros((D, D))
    check(lib().lc_gw_mstep(clustwidth, D, Ns, dptr(xs), dptr(xxs), C.byref(nu), C.byref(beta), dptr(m), dptr(iW),
                            C.byref(logdW), C.byref(fe), dptr(A), C.byref(cst)))
    return {"nu": nu.value, "beta": beta.value, "m": m, "iW": iW, "logdW": logdW.value, "fenergy": fe.value,
            "A": A, "eloglike_const": cst.value}


def ng_mstep(clustwidth, Ns, xs, xxs):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    xxs = np.ascontiguousarray(xxs, dtype=np.float64)
    D = xs.size
    nu, beta, logL, fe, cst = (C.c_double() for _ in range(5))
    m, L = np.zeros(D), np.zeros(D)
    check(lib().lc_ng_mstep(clustwidth, D, Ns, dptr(xs), dptr(xxs), C.byref(nu), C.byref(beta), dptr(m), dptr(L),
                            C.byref(logL), C.byref(fe), C.byref(cst)))
    return {"nu": nu.value, "beta": beta.value, "m": m, "L": L, "logL": logL.value, "fenergy": fe.value,
            "eloglike_const": cst.value}


def eg_mstep(obsmag, Ns, xs):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    D = xs.size
    a, logb, fe, cst = (C.c_double() for _ in range(4))
    ib = np.zeros(D)
    check(lib().lc_eg_mstep(obsmag, D, Ns, dptr(xs), C.byref(a), dptr(ib), C.byref(logb), C.byref(fe), C.byref(cst)))
    return {"a": a.value, "ib": ib, "logb": logb.value, "fenergy": fe.value, "eloglike_const": cst.value}


class TopicModel:
    """lc_tmodel: the result of learnSCM / learnMCM."""

    def __init__(self, handle, Ij, Nji):
        self._h = handle
        self.Ij = list(Ij)
        self.Nji = list(Nji)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().lc_tmodel_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dims(self):
        v = [C.c_int() for _ in range(6)]
        check(lib().lc_tmodel_dims(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("J", "Itot", "T", "K", "D", "Dt"), (x.value for x in v)))

    def qY(self):
        d = self.dims()
        out = []
        for j, I in enumerate(self.Ij):
            q = np.zeros((I, d["T"]))
            check(lib().lc_tmodel_get_qy(self._h, j, dptr(q)))
            out.append(q)
        return out

    def qZ(self):
        K = self.dims()["K"]
        allq = np.zeros((int(sum(self.Nji)), K))
        if allq.size:
            check(lib().lc_tmodel_get_qz_all(self._h, dptr(allq)))  # one transfer, then views per document
        out, doc, o = [], 0, 0
        for I in self.Ij:
            g = []
            for _ in range(I):
                g.append(allq[o:o + self.Nji[doc]])
                o += self.Nji[doc]
                doc += 1
            out.append(g)
        return out

    def weights(self, level, idx):
        d = self.dims()
        n = d["T"] if level == 0 else d["K"]
        e, nk = np.zeros(n), np.zeros(n)
        check(lib().lc_tmodel_weights(self._h, level, idx, dptr(e), dptr(nk)))
        return e, nk

    def cluster(self, level, idx):
        d = self.dims()
        D = d["D"] if level == 0 else d["Dt"]
        N, nu, beta, logdW, fe = (C.c_double() for _ in range(5))
        mean, cov, iW = np.zeros(D), np.zeros((D, D)), np.zeros((D, D))
        check(lib().lc_tmodel_cluster(self._h, level, idx, C.byref(N), dptr(mean), dptr(cov), C.byref(nu),
                                      C.byref(beta), dptr(iW), C.byref(logdW), C.byref(fe)))
        return {"N": N.value, "mean": mean, "cov": cov, "nu": nu.value, "beta": beta.value, "iW": iW,
                "logdW": logdW.value, "fenergy": fe.value}

    def rounds(self):
        n = C.c_int()
        check(lib().lc_tmodel_rounds(self._h, C.byref(n)))
        out = []
        for r in range(n.value):
            T, K, ni = C.c_int(), C.c_int(), C.c_int()
            check(lib().lc_tmodel_round(self._h, r, C.byref(T), C.byref(K), C.byref(ni), None, 0))
            F = np.zeros(ni.value)
            check(lib().lc_tmodel_round(self._h, r, None, None, None, dptr(F), ni.value))
            out.append((T.value, K.value, F.tolist()))
        return out


def learn_topic(X, W=None, qY0=None, prior_t=1.0, prior_k=1.0, maxT=100, maxK=-1, verbose=False, nthreads=1,
                device=0, allreduce=None, stream=None):
    """lc_learn_topic: X is a list (groups) of lists (documents) of (N_ji, D) arrays; W (MCM) a list of (I_j, Dt)
    arrays; qY0 an optional list of (I_j, maxT) initial assignments.  Returns (F, TopicModel).
    allreduce (one process per GPU, whole groups per rank): fn(device_ptr, count, stream) summing in place across
    ranks, e.g. libcluster_amd.dist.make_device_hook(device); stream: the HIP stream to work on."""
    Xs = [[np.ascontiguousarray(x, dtype=np.float64) for x in Xj] for Xj in X]
    J = len(Xs)
    docs = [x for Xj in Xs for x in Xj]
    if not docs:
        raise ValueError("need at least one document")
    D = docs[0].shape[1]
    for x in docs:
        if x.ndim != 2 or x.shape[1] != D:
            raise ValueError("X dimensions are inconsistent between documents!")
    Ij = (C.c_int * J)(*[len(Xj) for Xj in Xs])
    ptrs = (c_double_p * len(docs))(*[dptr(x) for x in docs])
    Nji = (C.c_int64 * len(docs))(*[x.shape[0] for x in docs])
    Wp, Dt, Ws = None, 0, None
    if W is not None:
        Ws = [np.ascontiguousarray(w, dtype=np.float64) for w in W]
        if len(Ws) != J:  # mcluster.cpp:548-549
            raise ValueError("W and X need to have the same number of groups!")
        for j in range(J):
            if Ws[j].shape[0] != len(Xs[j]):  # mcluster.cpp:553-555
                raise ValueError("W and X need to have the same number of 'docs'!")
        Dt = Ws[0].shape[1]
        Wp = (c_double_p * J)(*[dptr(w) for w in Ws])
    Qp, Qs = None, None
    if qY0 is not None:
        Qs = [np.ascontiguousarray(q, dtype=np.float64) for q in qY0]
        for j in range(J):
            if Qs[j].shape != (len(Xs[j]), maxT):
                raise ValueError("qY0[j] must be (I_j, maxT)")
        Qp = (c_double_p * J)(*[dptr(q) for q in Qs])
    mh, F = C.c_void_p(), C.c_double()
    cb = C.cast(None, ALLREDUCE_FN)
    if allreduce is not None:
        def tramp(user, buf, count, strm):
            try:
                allreduce(buf or 0, count, strm or 0)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        cb = ALLREDUCE_FN(tramp)
    check(lib().lc_learn_topic_dist(J, Ij, ptrs, Nji, D, D, 1, Wp, Dt, Qp, prior_t, prior_k, maxT, maxK, int(verbose),
                                    nthreads, device, C.c_void_p(stream) if stream else None, cb, None, C.byref(mh),
                                    C.byref(F)))
    return F.value, TopicModel(mh, [len(Xj) for Xj in Xs], [x.shape[0] for x in docs])
