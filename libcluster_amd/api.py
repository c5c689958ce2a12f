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
