"""learnVDP / learnBGMM / learnGMC over the C-ABI, with the return shape of the
reference's Python binding (python/libclusterpy.cpp:135-213:
``(F, qZ, weights, means, covariances)``).  Argument meaning and error
behaviour follow include/libcluster.h:177-186, 218-227, 356-366."""
from __future__ import annotations

import numpy as np

from . import capi


def _result(F, model, rows, grouped):
    J, K, D = model.dims()
    qZ = [model.qz(j, rows[j]) for j in range(J)]
    w = [np.exp(model.weights(j)[0]) for j in range(J)]
    cl = [model.cluster(k) for k in range(K)]
    # ExpGamma has no mean / covariance: its slot carries getrate(), covs is None per cluster
    means = [c["mean"] if "mean" in c else c["rate"] for c in cl]
    covs = [c.get("cov") for c in cl]
    info = {"K": K, "N": [c["N"] for c in cl], "rounds": model.rounds(), "clusters": cl,
            "Elogweight": [model.weights(j)[0] for j in range(J)]}
    model.close()
    if grouped:
        return F, qZ, w, means, covs, info
    return F, qZ[0], w[0], means, covs, info


def learnVDP(X, prior=1.0, maxclusters=-1, verbose=False, nthreads=1, concentration=1.0, device=0):
    """include/libcluster.h:177-186.  Returns (F, qZ, weights, means, covs, info)."""
    F, m, rows = capi.learn(capi.ALGO_VDP, np.asarray(X, dtype=np.float64), concentration, prior, maxclusters,
                            False, verbose, nthreads, device)
    return _result(F, m, rows, False)


def learnBGMM(X, prior=1.0, maxclusters=-1, verbose=False, nthreads=1, alpha=1.0, device=0):
    """include/libcluster.h:218-227."""
    F, m, rows = capi.learn(capi.ALGO_BGMM, np.asarray(X, dtype=np.float64), alpha, prior, maxclusters, False,
                            verbose, nthreads, device)
    return _result(F, m, rows, False)


def learnGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1, device=0):
    """include/libcluster.h:356-366.  X is a list of (N_j, D) arrays."""
    F, m, rows = capi.learn(capi.ALGO_GMC, [np.asarray(x, dtype=np.float64) for x in X], 1.0, prior, maxclusters,
                            sparse, verbose, nthreads, device)
    return _result(F, m, rows, True)


def learnSGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1, device=0):
    """include/libcluster.h:409-419 (symmetric GMC: one Dirichlet per group).  X is a list of (N_j, D) arrays."""
    F, m, rows = capi.learn(capi.ALGO_SGMC, [np.asarray(x, dtype=np.float64) for x in X], 1.0, prior, maxclusters,
                            sparse, verbose, nthreads, device)
    return _result(F, m, rows, True)


def learnDGMM(X, prior=1.0, maxclusters=-1, verbose=False, nthreads=1, alpha=1.0, device=0):
    """include/libcluster.h:262-271 (diagonal Gaussians, NormGamma).  covs are the D-vectors getcov() returns."""
    F, m, rows = capi.learn(capi.ALGO_DGMM, np.asarray(X, dtype=np.float64), alpha, prior, maxclusters, False,
                            verbose, nthreads, device)
    return _result(F, m, rows, False)


def learnBEMM(X, prior=1.0, maxclusters=-1, verbose=False, nthreads=1, alpha=1.0, device=0):
    """include/libcluster.h:306-315 (exponential mixture).  ValueError if X has a negative entry; `means` holds
    the clusters' getrate(), covs are None."""
    F, m, rows = capi.learn(capi.ALGO_BEMM, np.asarray(X, dtype=np.float64), alpha, prior, maxclusters, False,
                            verbose, nthreads, device)
    return _result(F, m, rows, False)


def learnDGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1, device=0):
    """include/libcluster.h:462-472.  X is a list of (N_j, D) arrays."""
    F, m, rows = capi.learn(capi.ALGO_DGMC, [np.asarray(x, dtype=np.float64) for x in X], 1.0, prior, maxclusters,
                            sparse, verbose, nthreads, device)
    return _result(F, m, rows, True)


def learnEGMC(X, prior=1.0, maxclusters=-1, sparse=False, verbose=False, nthreads=1, device=0):
    """include/libcluster.h:513-523.  X is a list of non-negative (N_j, D) arrays."""
    F, m, rows = capi.learn(capi.ALGO_EGMC, [np.asarray(x, dtype=np.float64) for x in X], 1.0, prior, maxclusters,
                            sparse, verbose, nthreads, device)
    return _result(F, m, rows, True)


def _topic_result(F, m, mcm):
    d = m.dims()
    qY, qZ = m.qY(), m.qZ()
    wj = [np.exp(m.weights(0, j)[0]) for j in range(d["J"])]
    wt = [np.exp(m.weights(1, t)[0]) for t in range(d["T"])]
    ck = [m.cluster(0, k) for k in range(d["K"])]
    ct = [m.cluster(1, t) for t in range(d["T"])] if mcm else []
    info = {"T": d["T"], "K": d["K"], "rounds": m.rounds(), "clusters_k": ck, "clusters_t": ct,
            "Elogweight_j": [m.weights(0, j)[0] for j in range(d["J"])],
            "Elogweight_t": [m.weights(1, t)[0] for t in range(d["T"])]}
    m.close()
    if mcm:  # python/libclusterpy.cpp:305-307
        return (F, qY, qZ, wj, wt, [c["mean"] for c in ct], [c["mean"] for c in ck], [c["cov"] for c in ct],
                [c["cov"] for c in ck], info)
    return F, qY, qZ, wj, wt, [c["mean"] for c in ck], [c["cov"] for c in ck], info  # libclusterpy.cpp:270-271


def learnSCM(X, dirprior=1.0, gausprior=1.0, trunc=100, maxclusters=-1, verbose=False, nthreads=1, qY0=None, device=0):
    """include/libcluster.h:583-596, python/libclusterpy.cpp:244-272.  X: list (groups) of lists (documents) of
    (N_ji, D) arrays.  Returns (F, qY, qZ, weights_j, weights_t, means, covs, info).  qY0 (additive): initial
    (I_j, trunc) top-level assignments instead of the reference's std::rand() start."""
    F, m = capi.learn_topic(X, None, qY0, dirprior, gausprior, trunc, maxclusters, verbose, nthreads, device)
    return _topic_result(F, m, False)


def learnMCM(W, X, gausprior_t=1.0, gausprior_k=1.0, trunc=100, maxclusters=-1, verbose=False, nthreads=1, qY0=None,
             device=0):
    """include/libcluster.h:661-676, python/libclusterpy.cpp:276-308.  W: list of (I_j, Dt) document observations.
    Returns (F, qY, qZ, weights_j, weights_t, means_t, means_k, covs_t, covs_k, info)."""
    F, m = capi.learn_topic(X, W, qY0, gausprior_t, gausprior_k, trunc, maxclusters, verbose, nthreads, device)
    return _topic_result(F, m, True)
