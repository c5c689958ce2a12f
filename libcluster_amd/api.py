"""The reference's Python interface (python/libclusterpy.cpp, python/libclusterpy.h) over the C-ABI.

Same function names, keyword names (``prior``, ``maxclusters``, ``sparse``, ``verbose``, ``threads``, ``dirprior``,
``gausprior``, ``gausprior_t``, ``gausprior_k``, ``trunc``), defaults and return tuples as the Boost.Python module
(libclusterpy.h:300-470, libclusterpy.cpp:135-308), including its shapes: weights as (K, 1) arrays, means as a list
of (1, D) arrays, covariances as a list of (D, D) arrays -- so ``f, qZ, w, mu, cov = lc.learnVDP(X)`` of
python/testapi.py runs unchanged.  Like that module the prior arguments pass through ``float`` (C single
precision, libclusterpy.h:74-135).

Additive keyword-only extras: ``return_info=True`` appends a dict (rounds with every free energy, cluster
posteriors, E[log weights]), ``device`` picks the GPU, ``concentration`` / ``alpha`` set the weight prior the C++ API
takes through its ``weights`` argument, ``qY0`` gives learnSCM / learnMCM a reproducible start, ``nthreads`` is an
alias of ``threads``.  learnDGMM / learnBEMM / learnDGMC / learnEGMC are not in the reference's Python module
(they are in its C++ API, include/libcluster.h:262-315, 462-523) and follow the same conventions."""
from __future__ import annotations

import os

import numpy as np

from . import capi


def _f32(v):  # `const float clusterprior` of the Boost.Python wrappers
    return float(np.float32(v))


def _threads(threads, nthreads):
    if nthreads is not None:
        return int(nthreads)
    if threads is not None:
        return int(threads)
    return max(1, min(32, os.cpu_count() or 1))  # omp_get_max_threads() in the reference


def _result(F, model, rows, grouped, return_info):
    J, K, D = model.dims()
    qZ = model.qz_all(rows)
    w = [np.exp(model.weights(j)[0]).reshape(-1, 1) for j in range(J)]  # ArrayXd -> (K, 1)
    cl = [model.cluster(k) for k in range(K)]
    # ExpGamma has no mean / covariance: its slot carries getrate(), covs is None per cluster
    means = [(c["mean"] if "mean" in c else c["rate"]).reshape(1, -1) for c in cl]  # RowVectorXd -> (1, D)
    covs = [c.get("cov") if c.get("cov") is None or c["cov"].ndim == 2 else c["cov"].reshape(1, -1) for c in cl]
    info = {"K": K, "N": [c["N"] for c in cl], "rounds": model.rounds(), "clusters": cl,
            "Elogweight": [model.weights(j)[0] for j in range(J)]}
    model.close()
    out = (F, qZ, w, means, covs) if grouped else (F, qZ[0], w[0], means, covs)
    return out + (info,) if return_info else out


def _flat(algo, X, wprior, prior, maxclusters, verbose, threads, nthreads, device, return_info):
    F, m, rows = capi.learn(algo, np.asarray(X, dtype=np.float64), wprior, _f32(prior), maxclusters, False, verbose,
                            _threads(threads, nthreads), device)
    return _result(F, m, rows, False, return_info)


def _grouped(algo, X, prior, maxclusters, sparse, verbose, threads, nthreads, device, return_info):
    F, m, rows = capi.learn(algo, [np.asarray(x, dtype=np.float64) for x in X], 1.0, _f32(prior), maxclusters,
                            sparse, verbose, _threads(threads, nthreads), device)
    return _result(F, m, rows, True, return_info)


def learnVDP(X, prior=1.0, maxclusters=-1, verbose=False, threads=None, *, nthreads=None, concentration=1.0,
             device=0, return_info=False):
    """libclusterpy.h:300-320 / include/libcluster.h:177-186.  Returns (f, qZ, w, mu, cov)."""
    return _flat(capi.ALGO_VDP, X, concentration, prior, maxclusters, verbose, threads, nthreads, device, return_info)


def learnBGMM(X, prior=1.0, maxclusters=-1, verbose=False, threads=None, *, nthreads=None, alpha=1.0, device=0,
              return_info=False):
    """libclusterpy.h:322-342 / include/libcluster.h:218-227.  Returns (f, qZ, w, mu, cov)."""
    return _flat(capi.ALGO_BGMM, X, alpha, prior, maxclusters, verbose, threads, nthreads, device, return_info)


def learnGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, threads=None, *, nthreads=None, device=0,
             return_info=False):
    """libclusterpy.h:344-368 / include/libcluster.h:356-366.  X: list of (N_j, D) arrays.
    Returns (f, qZ, w, mu, cov) with qZ and w lists over the groups."""
    return _grouped(capi.ALGO_GMC, X, prior, maxclusters, sparse, verbose, threads, nthreads, device, return_info)


def learnSGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, threads=None, *, nthreads=None, device=0,
              return_info=False):
    """libclusterpy.h:370-395 / include/libcluster.h:409-419 (one Dirichlet per group)."""
    return _grouped(capi.ALGO_SGMC, X, prior, maxclusters, sparse, verbose, threads, nthreads, device, return_info)


def learnDGMM(X, prior=1.0, maxclusters=-1, verbose=False, threads=None, *, nthreads=None, alpha=1.0, device=0,
              return_info=False):
    """include/libcluster.h:262-271 (diagonal Gaussians, NormGamma).  cov holds the (1, D) vectors getcov() returns."""
    return _flat(capi.ALGO_DGMM, X, alpha, prior, maxclusters, verbose, threads, nthreads, device, return_info)


def learnBEMM(X, prior=1.0, maxclusters=-1, verbose=False, threads=None, *, nthreads=None, alpha=1.0, device=0,
              return_info=False):
    """include/libcluster.h:306-315 (exponential mixture).  ValueError if X has a negative entry; mu holds the
    clusters' getrate(), cov is a list of None."""
    return _flat(capi.ALGO_BEMM, X, alpha, prior, maxclusters, verbose, threads, nthreads, device, return_info)


def learnDGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, threads=None, *, nthreads=None, device=0,
              return_info=False):
    """include/libcluster.h:462-472.  X is a list of (N_j, D) arrays."""
    return _grouped(capi.ALGO_DGMC, X, prior, maxclusters, sparse, verbose, threads, nthreads, device, return_info)


def learnEGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, threads=None, *, nthreads=None, device=0,
              return_info=False):
    """include/libcluster.h:513-523.  X is a list of non-negative (N_j, D) arrays."""
    return _grouped(capi.ALGO_EGMC, X, prior, maxclusters, sparse, verbose, threads, nthreads, device, return_info)


def _topic_result(F, m, mcm, return_info):
    d = m.dims()
    qY, qZ = m.qY(), m.qZ()
    wj = [np.exp(m.weights(0, j)[0]).reshape(-1, 1) for j in range(d["J"])]
    wt = [np.exp(m.weights(1, t)[0]).reshape(-1, 1) for t in range(d["T"])]
    ck = [m.cluster(0, k) for k in range(d["K"])]
    ct = [m.cluster(1, t) for t in range(d["T"])] if mcm else []
    info = {"T": d["T"], "K": d["K"], "rounds": m.rounds(), "clusters_k": ck, "clusters_t": ct,
            "Elogweight_j": [m.weights(0, j)[0] for j in range(d["J"])],
            "Elogweight_t": [m.weights(1, t)[0] for t in range(d["T"])]}
    m.close()
    mean = lambda cs: [c["mean"].reshape(1, -1) for c in cs]  # noqa: E731
    cov = lambda cs: [c["cov"] for c in cs]  # noqa: E731
    if mcm:  # libclusterpy.cpp:305-307
        out = (F, qY, qZ, wj, wt, mean(ct), mean(ck), cov(ct), cov(ck))
    else:    # libclusterpy.cpp:270-271
        out = (F, qY, qZ, wj, wt, mean(ck), cov(ck))
    return out + (info,) if return_info else out


def learnSCM(X, dirprior=1.0, gausprior=1.0, trunc=100, maxclusters=-1, verbose=False, threads=None, *,
             nthreads=None, qY0=None, device=0, return_info=False):
    """libclusterpy.h:397-425 / include/libcluster.h:583-596.  X: list (groups) of lists (documents) of (N_ji, D)
    arrays.  Returns (f, qY, qZ, w_j, w_t, mu, cov).  qY0: initial (I_j, trunc) top-level assignments instead of the
    reference's std::rand() start."""
    F, m = capi.learn_topic(X, None, qY0, _f32(dirprior), _f32(gausprior), int(trunc), maxclusters, verbose,
                            _threads(threads, nthreads), device)
    return _topic_result(F, m, False, return_info)


def learnMCM(W, X, gausprior_t=1.0, gausprior_k=1.0, trunc=100, maxclusters=-1, verbose=False, threads=None, *,
             nthreads=None, qY0=None, device=0, return_info=False):
    """libclusterpy.h:427-458 / include/libcluster.h:661-676.  W: list of (I_j, D_t) document observations.
    Returns (f, qY, qZ, w_j, w_t, mu_t, mu_k, cov_t, cov_k)."""
    F, m = capi.learn_topic(X, W, qY0, _f32(gausprior_t), _f32(gausprior_k), int(trunc), maxclusters, verbose,
                            _threads(threads, nthreads), device)
    return _topic_result(F, m, True, return_info)
