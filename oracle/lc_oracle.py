"""CPU oracle for the libcluster variational E-step / suff-stat hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``libcluster_amd/``,
the C-ABI library) may import, call or link this file.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker.

PARITY UNPINNED: the reference (dsteinberg/libcluster) cannot be compiled in
this image (every translation unit needs Eigen 3 and Boost.Math, neither is
installed and there is no network) and its tests hold no golden values or
assertions (test/cluster_test.cpp:38-69 only prints).  This file is a numpy /
scipy *restatement* of the reference arithmetic, line by line, each function
citing the reference file:line it follows.  What pins it instead:
  * ``scikit-learn``'s ``BayesianGaussianMixture`` private E-step, which
    implements the same Bishop PRML 10.2 expectations
    (tests/test_oracle.py::test_eloglike_matches_sklearn);
  * ``scipy.special.digamma/gammaln`` for the special functions Boost.Math
    provides in the reference;
  * the reference's own test data (test/testdata.h, re-typed as
    tests/golden/xcat.json) with end-to-end traces committed as fixtures;
  * an independent textbook derivation of the free energy (negative ELBO from
    Bishop PRML 10.71-10.77 and the Beta / Gamma / Normal-Gamma KL divergences,
    written without any of this file's code): equal to ``fenergy`` for all three
    weight and all three cluster families (tests/test_oracle_elbo.py).

All arithmetic is IEEE double, like the reference.  Matrices are numpy
row-major ``(N, D)`` arrays; groups are python lists of such arrays.
"""
from __future__ import annotations

import math
import sys

import numpy as np
from scipy.special import digamma, gammaln

# --- constants: include/libcluster.h:122-127 (note the *float* literals) ---
PRIORVAL = 1.0
TRUNC = 100
SPLITITER = 15
CONVERGE = float(np.float32(1e-5))
FENGYDEL = CONVERGE / 10
ZEROCUTOFF = float(np.float32(0.1))
# include/distributions.h:39-43
BETAPRIOR = 1.0
NUPRIOR = 1.0
ALPHA1PRIOR = 1.0
ALPHA2PRIOR = 1.0
# src/probutils.cpp:39-40
EIGCONTHRESH = float(np.float32(1.0e-8))
MAXITER = 100


# ---------------------------------------------------------------------------
# probutils
# ---------------------------------------------------------------------------

def logdet(A):
    """ln det(A) for SPD A -- src/probutils.cpp:189-202 (LDLT diag there)."""
    A = np.asarray(A, dtype=np.float64)
    if A.shape[0] != A.shape[1]:
        raise ValueError("Matrix A must be square!")
    try:
        L = np.linalg.cholesky(A)
    except np.linalg.LinAlgError as e:  # reference: domain_error
        raise ArithmeticError("Matrix A is not positive definite.") from e
    return 2.0 * float(np.sum(np.log(np.diag(L))))


def mahaldist(X, mu, A):
    """(x-mu)^T A^-1 (x-mu) per row -- src/probutils.cpp:113-138."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    mu = np.asarray(mu, dtype=np.float64).reshape(-1)
    A = np.asarray(A, dtype=np.float64)
    if X.shape[1] != mu.shape[0] or X.shape[1] != A.shape[1]:
        raise ValueError("Arguments do not have the same dimensionality")
    if A.shape[0] != A.shape[1]:
        raise ValueError("Matrix A must be square!")
    try:
        L = np.linalg.cholesky(A)
    except np.linalg.LinAlgError as e:
        raise ValueError("Matrix A is not positive definite") from e
    X_mu = (X - mu).T  # D x N, probutils.cpp:135
    sol = np.linalg.solve(L.T, np.linalg.solve(L, X_mu))  # A^-1 X_mu, :136
    return np.sum(X_mu * sol, axis=0)


def logsumexp(X):
    """Row-wise log-sum-exp -- src/probutils.cpp:141-150."""
    X = np.asarray(X, dtype=np.float64)
    if X.shape[0] == 0:
        return np.zeros(0)
    mx = X.max(axis=1)
    se = np.exp(X - mx[:, None]).sum(axis=1)
    return np.log(se) + mx


def eigpower(A):
    """Power method principal eigenpair -- src/probutils.cpp:153-186."""
    A = np.asarray(A, dtype=np.float64)
    if A.shape[0] != A.shape[1]:
        raise ValueError("Matrix A must be square!")
    n = A.shape[0]
    if n == 1:
        return float(A[0, 0]), np.ones(1)
    v = np.linspace(-1.0, 1.0, n)
    eigval = float(np.linalg.norm(v))
    eigvec = v / eigval
    vdist = math.inf
    i = 0
    while vdist > EIGCONTHRESH and i < MAXITER:
        oeigvec = eigvec
        v = A @ oeigvec
        eigval = float(np.linalg.norm(v))
        eigvec = v / eigval
        vdist = float(np.linalg.norm(eigvec - oeigvec))
        i += 1
    return eigval, eigvec


def std_sort(items, less):
    """``std::sort(first, last, comp)`` as libstdc++ implements it (bits/stl_algo.h: introsort = median-of-three
    quicksort down to ranges of 16, then one insertion sort; GCC 4 ... 14 are identical here).  The reference sorts
    the stick-breaking order (distributions.cpp:146) and the split candidates (cluster.cpp:418, scluster.cpp:331,
    mcluster.cpp:360) with it; std::sort is not stable, so with TIES and more than 16 elements the resulting order --
    and with it E[log pi] of the tied clusters -- depends on this exact algorithm.  ``less(a, b)`` is the comparator.
    libstdc++ is a dependency outside /root/reference; its published algorithm is restated here."""
    a = list(items)
    n = len(a)
    if n == 0:
        return a

    def unguarded_linear_insert(last):
        val = a[last]
        nxt = last - 1
        while less(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                unguarded_linear_insert(i)

    def move_median_to_first(result, ia, ib, ic):
        if less(a[ia], a[ib]):
            if less(a[ib], a[ic]):
                pick = ib
            elif less(a[ia], a[ic]):
                pick = ic
            else:
                pick = ia
        elif less(a[ia], a[ic]):
            pick = ia
        elif less(a[ib], a[ic]):
            pick = ic
        else:
            pick = ib
        a[result], a[pick] = a[pick], a[result]

    def unguarded_partition(first, last, pivot):
        while True:
            while less(a[first], a[pivot]):
                first += 1
            last -= 1
            while less(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            a[first], a[last] = a[last], a[first]
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                raise NotImplementedError("std::sort fell back to heapsort (depth limit); not restated")
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    introsort_loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            unguarded_linear_insert(i)
    else:
        insertion_sort(0, n)
    return a



def enumdims(D):
    """1..D -- src/distributions.cpp:66-76."""
    return np.arange(1, D + 1, dtype=np.float64) if D > 1 else np.ones(1)


# ---------------------------------------------------------------------------
# weight distributions
# ---------------------------------------------------------------------------

class Dirichlet:
    """src/distributions.cpp:222-266, include/distributions.h:163-189."""

    def __init__(self, alpha=ALPHA1PRIOR):
        if alpha <= 0:
            raise ValueError("Alpha prior must be > 0!")
        self.alpha_p = float(alpha)
        self.alpha = np.full(1, self.alpha_p)
        self.E_logpi = np.zeros(1)
        self.Nk = np.zeros(1)  # WeightDist(), distributions.h:94

    def update(self, Nk):  # :242-256
        Nk = np.asarray(Nk, dtype=np.float64).reshape(-1)
        self.Nk = Nk.copy()
        self.alpha = self.alpha_p + Nk
        self.E_logpi = digamma(self.alpha) - digamma(self.alpha.sum())

    def Elogweight(self):
        return self.E_logpi

    def getNk(self):
        return self.Nk

    def fenergy(self):  # :259-266
        K = self.alpha.size
        return float(
            gammaln(self.alpha.sum())
            - (self.alpha_p - 1) * self.E_logpi.sum()
            + ((self.alpha - 1) * self.E_logpi - gammaln(self.alpha)).sum()
            - gammaln(K * self.alpha_p)
            + K * gammaln(self.alpha_p)
        )


class StickBreak:
    """src/distributions.cpp:83-179, include/distributions.h:103-140."""

    def __init__(self, concentration=ALPHA1PRIOR):
        if concentration <= 0:
            raise ValueError("Concentration parameter has to be > 0!")
        self.alpha1_p = float(concentration)
        self.alpha2_p = ALPHA2PRIOR
        self.alpha1 = np.full(1, self.alpha1_p)
        self.alpha2 = np.full(1, self.alpha2_p)
        self.E_logv = np.zeros(1)
        self.E_lognv = np.zeros(1)
        self.E_logpi = np.zeros(1)
        self.order = [0]
        self.Nk = np.zeros(1)
        # priorfcalc, :116-121
        self.F_p = float(
            gammaln(self.alpha1_p) + gammaln(self.alpha2_p)
            - gammaln(self.alpha1_p + self.alpha2_p)
        )

    def update(self, Nk):  # :124-168
        Nk = np.asarray(Nk, dtype=np.float64).reshape(-1)
        K = Nk.size
        self.Nk = Nk.copy()
        self.alpha1 = self.alpha1_p + Nk
        self.alpha2 = np.empty(K)
        self.E_logv = np.empty(K)
        self.E_lognv = np.empty(K)
        self.E_logpi = np.empty(K)
        # descending by size (:141-146) with std::sort's own (unstable) tie order, see std_sort
        self.order = std_sort(range(K), lambda i, j: Nk[i] > Nk[j])
        N = Nk.sum()
        cumNk = 0.0
        cumE_lognv = 0.0
        for k in self.order:
            cumNk += Nk[k]
            self.alpha2[k] = self.alpha2_p + (N - cumNk)
            psisum = digamma(self.alpha1[k] + self.alpha2[k])
            self.E_logv[k] = digamma(self.alpha1[k]) - psisum
            self.E_lognv[k] = digamma(self.alpha2[k]) - psisum
            self.E_logpi[k] = self.E_logv[k] + cumE_lognv
            cumE_lognv += self.E_lognv[k]

    def Elogweight(self):
        return self.E_logpi

    def getNk(self):
        return self.Nk

    def fenergy(self):  # :171-179
        K = self.alpha1.size
        return float(
            K * self.F_p
            + (
                gammaln(self.alpha1 + self.alpha2)
                - gammaln(self.alpha1)
                - gammaln(self.alpha2)
                + (self.alpha1 - self.alpha1_p) * self.E_logv
                + (self.alpha2 - self.alpha2_p) * self.E_lognv
            ).sum()
        )


class GDirichlet(StickBreak):
    """src/distributions.cpp:186-215 (default-constructed only)."""

    def __init__(self):
        super().__init__()

    def update(self, Nk):  # :186-196
        super().update(Nk)
        smallk = self.order[-1]
        self.E_logpi[smallk] = self.E_logpi[smallk] - self.E_logv[smallk]
        self.E_logv[smallk] = 0.0
        self.E_lognv[smallk] = 0.0

    def fenergy(self):  # :199-215
        K = len(self.order)
        Fpi = 0.0
        for k in self.order[: K - 1]:
            Fpi += (
                gammaln(self.alpha1[k] + self.alpha2[k])
                - gammaln(self.alpha1[k])
                - gammaln(self.alpha2[k])
                + (self.alpha1[k] - self.alpha1_p) * self.E_logv[k]
                + (self.alpha2[k] - self.alpha2_p) * self.E_lognv[k]
            )
        return float((K - 1) * self.F_p + Fpi)


# ---------------------------------------------------------------------------
# cluster distribution
# ---------------------------------------------------------------------------

class GaussWish:
    """src/distributions.cpp:273-399, include/distributions.h:279-337."""

    def __init__(self, clustwidth, D):
        if clustwidth <= 0:
            raise ValueError("clustwidth must be > 0!")
        self.D = int(D)
        self.prior = float(clustwidth)
        self.N = 0.0
        self.nu_p = float(D)
        self.beta_p = BETAPRIOR
        self.m_p = np.zeros(D)
        self.iW_p = self.nu_p * self.prior * np.eye(D)  # :286
        self.logdW_p = -logdet(self.iW_p)
        self.F_p = float(gammaln((self.nu_p + 1 - enumdims(D)) / 2).sum())  # :294
        self.clearobs()

    def clearobs(self):  # :340-353
        self.nu = self.nu_p
        self.beta = self.beta_p
        self.m = self.m_p.copy()
        self.iW = self.iW_p.copy()
        self.logdW = self.logdW_p
        self.N_s = 0.0
        self.x_s = np.zeros(self.D)
        self.xx_s = np.zeros((self.D, self.D))

    def addobs(self, qZk, X):  # :301-313
        X = np.asarray(X, dtype=np.float64)
        qZk = np.asarray(qZk, dtype=np.float64).reshape(-1)
        if X.shape[1] != self.D:
            raise ValueError("Mismatched dims. of cluster params and obs.!")
        if qZk.shape[0] != X.shape[0]:
            raise ValueError("qZk and X ar not the same length!")
        qZkX = qZk[:, None] * X
        self.N_s += qZk.sum()
        self.x_s = self.x_s + qZkX.sum(axis=0)
        self.xx_s = self.xx_s + qZkX.T @ X

    def addstats(self, N_s, x_s, xx_s):
        """Not in the reference: inject already-reduced statistics (what the
        device kernels return); equals a sequence of addobs calls."""
        self.N_s += float(N_s)
        self.x_s = self.x_s + np.asarray(x_s, dtype=np.float64)
        self.xx_s = self.xx_s + np.asarray(xx_s, dtype=np.float64)

    def update(self):  # :316-337
        xk = np.zeros(self.D)
        if self.N_s > 0:
            xk = self.x_s / self.N_s
        Sk = self.xx_s - np.outer(xk, self.x_s)
        xk_m = xk - self.m_p
        self.N = self.N_s
        self.nu = self.nu_p + self.N
        self.beta = self.beta_p + self.N
        self.m = (self.beta_p * self.m_p + self.x_s) / self.beta
        self.iW = self.iW_p + Sk + (self.beta_p * self.N / self.beta) * np.outer(xk_m, xk_m)
        try:
            self.logdW = -logdet(self.iW)
        except ArithmeticError as e:
            raise RuntimeError("Calc log(det(W)): " + str(e)) from e

    def sumpsi(self):
        return float(digamma((self.nu + 1 - enumdims(self.D)) / 2).sum())

    def Eloglike(self, X):  # :356-370
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        return 0.5 * (
            self.sumpsi()
            + self.logdW
            - self.D * (1 / self.beta + math.log(math.pi))
            - self.nu * mahaldist(X, self.m, self.iW)
        )

    def splitobs(self, X):  # :373-385
        _, eigvec = eigpower(self.iW)
        X = np.atleast_2d(np.asarray(X, dtype=np.float64)).reshape(-1, self.D)
        return ((X - self.m) * eigvec).sum(axis=1) >= 0

    def fenergy(self):  # :388-399
        l = enumdims(self.D)
        sumpsi = float(digamma((self.nu + 1 - l) / 2).sum())
        tr = float(np.trace(np.linalg.solve(self.iW, self.iW_p)))
        maha = float(mahaldist(self.m[None, :], self.m_p, self.iW)[0])
        return float(
            self.F_p
            + (
                self.D * (self.beta_p / self.beta - 1 - self.nu - math.log(self.beta_p / self.beta))
                + self.nu * (tr + self.beta_p * maha)
                + self.nu_p * (self.logdW_p - self.logdW)
                + self.N * sumpsi
            )
            / 2
            - gammaln((self.nu + 1 - l) / 2).sum()
        )

    def getN(self):
        return self.N

    def getprior(self):
        return self.prior

    def getmean(self):
        return self.m

    def getcov(self):
        return self.iW / self.nu


class NormGamma:
    """Diagonal Gaussian clusters -- src/distributions.cpp:406-517, include/distributions.h:343-398."""

    def __init__(self, clustwidth, D):
        if clustwidth <= 0:
            raise ValueError("clustwidth must be > 0!")
        self.D = int(D)
        self.prior = float(clustwidth)
        self.N = 0.0
        self.nu_p = NUPRIOR
        self.beta_p = BETAPRIOR
        self.m_p = np.zeros(D)
        self.L_p = self.nu_p * self.prior * np.ones(D)  # :419
        self.logL_p = float(np.log(self.L_p).sum())
        self.clearobs()

    def clearobs(self):  # :467-480
        self.nu = self.nu_p
        self.beta = self.beta_p
        self.m = self.m_p.copy()
        self.L = self.L_p.copy()
        self.logL = self.logL_p
        self.N_s = 0.0
        self.x_s = np.zeros(self.D)
        self.xx_s = np.zeros(self.D)

    def addobs(self, qZk, X):  # :426-438
        X = np.asarray(X, dtype=np.float64)
        qZk = np.asarray(qZk, dtype=np.float64).reshape(-1)
        if X.shape[1] != self.D:
            raise ValueError("Mismatched dims. of cluster params and obs.!")
        if qZk.shape[0] != X.shape[0]:
            raise ValueError("qZk and X ar not the same length!")
        qZkX = qZk[:, None] * X
        self.N_s += qZk.sum()
        self.x_s = self.x_s + qZkX.sum(axis=0)
        self.xx_s = self.xx_s + (qZkX * X).sum(axis=0)

    def addstats(self, N_s, x_s, xx_s):
        self.N_s += float(N_s)
        self.x_s = self.x_s + np.asarray(x_s, dtype=np.float64)
        self.xx_s = self.xx_s + np.asarray(xx_s, dtype=np.float64)

    def update(self):  # :441-464
        xk = np.zeros(self.D)
        Sk = np.zeros(self.D)
        if self.N_s > 0:
            xk = self.x_s / self.N_s
            Sk = self.xx_s - self.x_s ** 2 / self.N_s
        self.N = self.N_s
        self.beta = self.beta_p + self.N
        self.nu = self.nu_p + self.N / 2
        self.m = (self.beta_p * self.m_p + self.x_s) / self.beta
        self.L = self.L_p + Sk / 2 + (self.beta_p * self.N / (2 * self.beta)) * (xk - self.m_p) ** 2
        if (self.L <= 0).any():
            raise ValueError("Calc log(L): Variance is zero or less!")
        self.logL = float(np.log(self.L).sum())

    def Eloglike(self, X):  # :483-492
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        Xmdist = ((X - self.m) ** 2) @ (1.0 / self.L)
        return 0.5 * (self.D * (digamma(self.nu) - math.log(2 * math.pi) - 1 / self.beta)
                      - self.logL - self.nu * Xmdist)

    def splitobs(self, X):  # :495-505
        X = np.atleast_2d(np.asarray(X, dtype=np.float64)).reshape(-1, self.D)
        ax = int(np.argmax(self.L))
        return (X[:, ax] - self.m[ax]) >= 0

    def fenergy(self):  # :508-517 -- note D/2 is INTEGER division there (D is unsigned int)
        iL = 1.0 / self.L
        D = self.D
        return float(
            D * (gammaln(self.nu_p) - gammaln(self.nu) + self.N * digamma(self.nu) / 2 - self.nu)
            + (D // 2) * (math.log(self.beta) - math.log(self.beta_p) - 1 + self.beta_p / self.beta)
            + self.beta_p * self.nu / 2 * (((self.m - self.m_p) ** 2) @ iL)
            + self.nu_p * (self.logL - self.logL_p)
            + self.nu * (self.L_p @ iL)
        )

    def getN(self):
        return self.N

    def getprior(self):
        return self.prior

    def getmean(self):
        return self.m

    def getcov(self):  # distributions.h:375 returns L*nu (sic: the expected variance would be L/nu); kept as is
        return self.L * self.nu


class ExpGamma:
    """Exponential clusters -- src/distributions.cpp:524-589, include/distributions.h:404-457."""

    def __init__(self, obsmag, D):
        self.D = int(D)
        self.prior = float(obsmag)
        self.N = 0.0
        self.a_p = 1.0  # APRIOR, distributions.h:43
        self.b_p = float(obsmag)
        self.clearobs()

    def clearobs(self):  # :555-565
        self.a = self.a_p
        self.ib = np.full(self.D, 1.0 / self.b_p)
        self.logb = self.D * math.log(self.b_p)
        self.N_s = 0.0
        self.x_s = np.zeros(self.D)

    def addobs(self, qZk, X):  # :533-542
        X = np.asarray(X, dtype=np.float64)
        qZk = np.asarray(qZk, dtype=np.float64).reshape(-1)
        if X.shape[1] != self.D:
            raise ValueError("Mismatched dims. of cluster params and obs.!")
        if qZk.shape[0] != X.shape[0]:
            raise ValueError("qZk and X ar not the same length!")
        self.N_s += qZk.sum()
        self.x_s = self.x_s + (qZk[:, None] * X).sum(axis=0)

    def addstats(self, N_s, x_s, xx_s=None):
        self.N_s += float(N_s)
        self.x_s = self.x_s + np.asarray(x_s, dtype=np.float64)

    def update(self):  # :545-552
        self.N = self.N_s
        self.a = self.a_p + self.N
        self.ib = 1.0 / (self.b_p + self.x_s)
        self.logb = -float(np.log(self.ib).sum())

    def Eloglike(self, X):  # :568-572
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        return self.D * digamma(self.a) - self.logb - self.a * (X @ self.ib)

    def splitobs(self, X):  # :575-581
        X = np.atleast_2d(np.asarray(X, dtype=np.float64)).reshape(-1, self.D)
        XdotL = X @ (self.a * self.ib)
        if XdotL.size == 0:
            return np.zeros(0, dtype=bool)
        return XdotL > (XdotL.sum() / XdotL.size)

    def fenergy(self):  # :584-589
        return float(
            self.D * ((self.a - self.a_p) * digamma(self.a) - self.a - self.a_p * math.log(self.b_p)
                      - gammaln(self.a) + gammaln(self.a_p))
            + self.b_p * self.a * self.ib.sum() + self.a_p * self.logb
        )

    def getN(self):
        return self.N

    def getprior(self):
        return self.prior

    def getrate(self):  # distributions.h: getrate() = a * ib
        return self.a * self.ib


# ---------------------------------------------------------------------------
# comutils
# ---------------------------------------------------------------------------

def partobs(X, Xpart):
    """src/comutils.cpp:56-72 -> (pidx, Xk)."""
    pidx = np.flatnonzero(Xpart)
    return pidx, np.asarray(X)[pidx, :].copy()


def auglabels(k, mapidx, Zsplit, qZ):
    """src/comutils.cpp:75-104."""
    Zsplit = np.asarray(Zsplit, dtype=bool)
    if Zsplit.size != mapidx.size:
        raise ValueError("map and split must be the same size!")
    N, K = qZ.shape
    qZaug = np.zeros((N, K + 1))
    qZaug[:, :K] = qZ
    rows = mapidx[Zsplit]
    qZaug[rows, K] = qZ[rows, k]
    qZaug[rows, k] = 0.0
    return qZaug


def anyempty(clusters):
    """src/comutils.h:114-123."""
    return any(c.getN() <= 1 for c in clusters)


# ---------------------------------------------------------------------------
# cluster.cpp
# ---------------------------------------------------------------------------

def _kful(K, sparse, Nk):
    """Active-cluster selection shared by updateSS/vbexpectation
    (src/cluster.cpp:63-70, 106-112)."""
    if not sparse:
        return np.arange(K) if K > 1 else np.zeros(1, dtype=int), np.zeros(0, dtype=int)
    mask = np.asarray(Nk) >= ZEROCUTOFF
    return np.flatnonzero(mask), np.flatnonzero(~mask)


def updateSS(Xj, qZj, clusters, sparse=False):
    """src/cluster.cpp:53-82."""
    K = qZj.shape[1]
    Njk = qZj.sum(axis=0)
    Kful, _ = _kful(K, sparse, Njk)
    for k in Kful:
        clusters[k].addobs(qZj[:, k], Xj)
    return Njk


def vbexpectation(Xj, weights, clusters, sparse=False):
    """src/cluster.cpp:91-138 -> (qZj, -sum(logZ))."""
    K = len(clusters)
    Nj = Xj.shape[0]
    E_logZ = weights.Elogweight()
    Kful, Kemp = _kful(K, sparse, weights.getNk())
    logqZj = np.empty((Nj, Kful.size))
    for i, k in enumerate(Kful):
        logqZj[:, i] = E_logZ[k] + clusters[k].Eloglike(Xj)
    logZzj = logsumexp(logqZj)
    qZj = np.zeros((Nj, K))
    for i, k in enumerate(Kful):
        qZj[:, k] = np.exp(logqZj[:, i] - logZzj)
    return qZj, -float(logZzj.sum())


def fenergy(weights, clusters, Fxz):
    """src/cluster.cpp:145-165."""
    Fw = 0.0
    for w in weights:
        Fw += w.fenergy()
    Fc = 0.0
    for c in clusters:
        Fc += c.fenergy()
    return Fc + Fw + Fxz


def vbem(X, qZ, weights, clusters, clusterprior, maxit=-1, sparse=False,
         verbose=False, wfactory=None, trace=None, cfactory=None):
    """src/cluster.cpp:177-239.  ``qZ``, ``weights``, ``clusters`` are lists
    mutated in place (the reference takes them by mutable reference).
    ``trace`` (optional list) receives F after every iteration."""
    J = len(X)
    K = qZ[0].shape[1]
    D = X[0].shape[1]
    while len(weights) < J:  # weights.resize(J, W()) :192
        weights.append(wfactory())
    while len(clusters) < K:  # clusters.resize(K, C(prior, D)) :193
        clusters.append((cfactory or GaussWish)(clusterprior, D))
    # (resize also shrinks)
    del weights[J:]
    del clusters[K:]

    F = sys.float_info.max
    i = 0
    while True:
        Fold = F
        for k in range(K):
            clusters[k].clearobs()
        for j in range(J):
            Njk = updateSS(X[j], qZ[j], clusters, sparse)
            weights[j].update(Njk)
        for k in range(K):
            clusters[k].update()
        Fz = 0.0
        for j in range(J):
            qZ[j], fz = vbexpectation(X[j], weights[j], clusters, sparse)
            Fz += fz
        F = fenergy(weights, clusters, Fz)
        if trace is not None:
            trace.append(F)
        if (F - Fold) / abs(Fold) > FENGYDEL:
            raise RuntimeError("Free energy increase!")
        if verbose:
            print("-", end="", flush=True)
        # while (|dF/F| > CONVERGE) && ((i++ < maxit) || (maxit < 0)) :235-236
        if not (abs((Fold - F) / Fold) > CONVERGE):
            break
        cont = (i < maxit) or (maxit < 0)
        i += 1
        if not cont:
            break
    return F


def prune_clusters(qZ, weights, clusters, verbose=False):
    """src/cluster.cpp:505-552."""
    K = len(clusters)
    Nk = np.array([c.getN() for c in clusters])
    empty = Nk < ZEROCUTOFF
    if not empty.any():
        return False
    if verbose:
        print("*", end="", flush=True)
    fidx = np.flatnonzero(~empty)
    for i in sorted(np.flatnonzero(empty), reverse=True):
        del clusters[i]
    for j in range(len(qZ)):
        qZ[j] = qZ[j][:, fidx].copy()
        weights[j].update(qZ[j].sum(axis=0))
    return True


def split_gr(X, weights, clusters, qZ, tally, F, maxclusters, sparse, verbose,
             wfactory, events=None, cfactory=None):
    """src/cluster.cpp:366-495."""
    J = len(X)
    K = len(clusters)
    if maxclusters >= 0 and K >= maxclusters:
        return False
    while len(tally) < K:
        tally.append(0)
    del tally[K:]

    Fk = np.array([c.fenergy() for c in clusters])
    for j in range(J):
        logpi = weights[j].Elogweight()
        for k in range(K):
            LL = float(qZ[j][:, k] @ (logpi[k] + clusters[k].Eloglike(X[j])))
            Fk[k] -= LL
    # greedcomp: tally ascending, then Fk descending (src/comutils.h:60-68)
    order = std_sort(range(K), lambda i, j: Fk[i] > Fk[j] if tally[i] == tally[j] else tally[i] < tally[j])
    if events is not None:
        events.append(("order", list(order), Fk.tolist()))

    for k in order:
        tally[k] += 1
        if clusters[k].getN() < 4:
            continue
        scount = 0
        Mtot = 0
        mapidx, Xk, qZref = [], [], []
        for j in range(J):
            idx, xk = partobs(X[j], qZ[j][:, k] > 0.5)
            mapidx.append(idx)
            Xk.append(xk)
            Mtot += xk.shape[0]
            splitk = clusters[k].splitobs(xk) if xk.shape[0] else np.zeros(0, dtype=bool)
            q = np.zeros((xk.shape[0], 2))
            q[:, 0] = splitk.astype(np.float64)
            q[:, 1] = (~splitk).astype(np.float64)
            qZref.append(q)
            scount += int(splitk.sum())
        if scount < 2 or scount > Mtot - 2:
            continue
        wspl, cspl = [], []
        vbem(Xk, qZref, wspl, cspl, clusters[0].getprior(), SPLITITER, sparse,
             wfactory=wfactory, cfactory=cfactory)
        if anyempty(cspl):
            continue
        qZaug = [auglabels(k, mapidx[j], qZref[j][:, 1] > 0.5, qZ[j]) for j in range(J)]
        Fsplit = vbem(X, qZaug, wspl, cspl, clusters[0].getprior(), 1, sparse,
                      wfactory=wfactory, cfactory=cfactory)
        if anyempty(cspl):
            continue
        if verbose:
            print("=", end="", flush=True)
        if events is not None:
            events.append(("candidate", int(k), float(Fsplit)))
        if Fsplit < F and abs((F - Fsplit) / F) > CONVERGE:
            for j in range(J):
                qZ[j] = qZaug[j]
            tally[k] = 0
            return True
    return False


def cluster(X, weights, clusters, clusterprior, maxclusters, sparse, verbose,
            wfactory, trace=None, events=None, cfactory=None):
    """src/cluster.cpp:564-629 -> (F, qZ)."""
    J = len(X)
    qZ = [np.ones((X[j].shape[0], 1)) for j in range(J)]
    tally = []
    issplit = True
    F = None
    while issplit:
        rtrace = [] if trace is not None else None
        F = vbem(X, qZ, weights, clusters, clusterprior, -1, sparse, verbose,
                 wfactory=wfactory, trace=rtrace, cfactory=cfactory)
        if trace is not None:
            trace.append((len(clusters), rtrace))
        prune_clusters(qZ, weights, clusters, verbose)
        if verbose:
            print("<", end="", flush=True)
        issplit = split_gr(X, weights, clusters, qZ, tally, F, maxclusters,
                           sparse, verbose, wfactory, events=events, cfactory=cfactory)
        if verbose:
            print(">")
    if verbose:
        print("Finished!")
        print("Number of clusters =", len(clusters))
        print("Free energy =", F)
    return F, qZ


def learnVDP(X, clusterprior=PRIORVAL, maxclusters=-1, verbose=False,
             weights=None, trace=None, events=None):
    """src/cluster.cpp:636-664 -> (F, qZ, weights, clusters)."""
    w = [weights if weights is not None else StickBreak()]
    clusters = []
    F, qZ = cluster([np.asarray(X, dtype=np.float64)], w, clusters, clusterprior,
                    maxclusters, False, verbose, StickBreak, trace, events)
    return F, qZ[0], w[0], clusters


def learnBGMM(X, clusterprior=PRIORVAL, maxclusters=-1, verbose=False,
              weights=None, trace=None, events=None):
    """src/cluster.cpp:667-695 -> (F, qZ, weights, clusters)."""
    w = [weights if weights is not None else Dirichlet()]
    clusters = []
    F, qZ = cluster([np.asarray(X, dtype=np.float64)], w, clusters, clusterprior,
                    maxclusters, False, verbose, Dirichlet, trace, events)
    return F, qZ[0], w[0], clusters


def learnGMC(X, clusterprior=PRIORVAL, maxclusters=-1, sparse=False,
             verbose=False, trace=None, events=None):
    """src/cluster.cpp:763-784 -> (F, qZ, weights, clusters)."""
    w, clusters = [], []
    Xl = [np.asarray(x, dtype=np.float64) for x in X]
    F, qZ = cluster(Xl, w, clusters, clusterprior, maxclusters, sparse, verbose,
                    GDirichlet, trace, events)
    return F, qZ, w, clusters


def learnSGMC(X, clusterprior=PRIORVAL, maxclusters=-1, sparse=False,
              verbose=False, trace=None, events=None):
    """src/cluster.cpp:787-807 -> (F, qZ, weights, clusters)."""
    w, clusters = [], []
    Xl = [np.asarray(x, dtype=np.float64) for x in X]
    F, qZ = cluster(Xl, w, clusters, clusterprior, maxclusters, sparse, verbose,
                    Dirichlet, trace, events)
    return F, qZ, w, clusters


def _nonneg(Xl):
    for x in Xl:
        if (x < 0).any():
            raise ValueError("X has to be in the range [0, inf)!")


def learnDGMM(X, clusterprior=PRIORVAL, maxclusters=-1, verbose=False, trace=None):
    """src/cluster.cpp:698-726 (Dirichlet weights, diagonal Gaussian clusters)."""
    w, clusters = [Dirichlet()], []
    F, qZ = cluster([np.asarray(X, dtype=np.float64)], w, clusters, clusterprior, maxclusters, False, verbose,
                    Dirichlet, trace, None, NormGamma)
    return F, qZ[0], w[0], clusters


def learnBEMM(X, clusterprior=PRIORVAL, maxclusters=-1, verbose=False, trace=None):
    """src/cluster.cpp:729-760 (Dirichlet weights, exponential clusters; X >= 0)."""
    Xa = np.asarray(X, dtype=np.float64)
    _nonneg([Xa])
    w, clusters = [Dirichlet()], []
    F, qZ = cluster([Xa], w, clusters, clusterprior, maxclusters, False, verbose, Dirichlet, trace, None, ExpGamma)
    return F, qZ[0], w[0], clusters


def learnDGMC(X, clusterprior=PRIORVAL, maxclusters=-1, sparse=False, verbose=False, trace=None):
    """src/cluster.cpp:810-831 (GDirichlet per group, diagonal Gaussian clusters)."""
    w, clusters = [], []
    Xl = [np.asarray(x, dtype=np.float64) for x in X]
    F, qZ = cluster(Xl, w, clusters, clusterprior, maxclusters, sparse, verbose, GDirichlet, trace, None, NormGamma)
    return F, qZ, w, clusters


def learnEGMC(X, clusterprior=PRIORVAL, maxclusters=-1, sparse=False, verbose=False, trace=None):
    """src/cluster.cpp:834-859 (GDirichlet per group, exponential clusters; X >= 0)."""
    Xl = [np.asarray(x, dtype=np.float64) for x in X]
    _nonneg(Xl)
    w, clusters = [], []
    F, qZ = cluster(Xl, w, clusters, clusterprior, maxclusters, sparse, verbose, GDirichlet, trace, None, ExpGamma)
    return F, qZ, w, clusters


# ---------------------------------------------------------------------------
# Simultaneous / multiple-source clustering models: src/scluster.cpp (learnSCM) and src/mcluster.cpp (learnMCM).
# X[j][i] is the (N_ji x D) matrix of "document" i of group j; qY[j] is (I_j x T); qZ[j][i] is (N_ji x K);
# W[j] (MCM only) is the (I_j x Dt) matrix of document-level observations.  One restatement serves both: the MCM
# is the SCM plus the top-level Gaussian clusters over W (mcluster.cpp differs from scluster.cpp only there, in
# the default-constructed weights_t, and in its guards for empty documents).
# ---------------------------------------------------------------------------

def random_qY(I, maxT, rng):
    """scluster.cpp:519-521 / mcluster.cpp:559-561: |U(-1,1)| rows, normalised.  The reference draws from Eigen's
    Random() (std::rand); here the generator is the caller's numpy Generator -- the learners take an explicit qY0
    for reproducible parity."""
    r = np.abs(rng.uniform(-1.0, 1.0, (I, maxT)))
    return np.exp(np.log(r) - np.log(r.sum(axis=1))[:, None])


def vbeY(qZj, weightsj, weights_t, qYshape, Wj=None, clusters_t=None):
    """scluster.cpp:50-85 / mcluster.cpp:49-92 -> (qYj, Fyz_j)."""
    T = len(weights_t)
    Ij = len(qZj)
    if Ij == 0:  # mcluster.cpp:64-65
        return np.zeros((0, T)), 0.0
    E_logwj = weightsj.Elogweight()
    Njik = np.stack([q.sum(axis=0) for q in qZj])
    like = np.empty((Ij, T))
    logq = np.empty((Ij, T))
    for t in range(T):
        like[:, t] = Njik @ weights_t[t].Elogweight()
        if clusters_t is None:
            logq[:, t] = E_logwj[t] + like[:, t]
        else:
            logq[:, t] = like[:, t] + E_logwj[t] + clusters_t[t].Eloglike(Wj)
    logZ = logsumexp(logq)
    qY = np.exp(logq - logZ[:, None])
    return qY, float(((qY * like).sum(axis=1) - logZ).sum())


def vbeZ(Xji, qYji, weights_t, clusters):
    """scluster.cpp:93-124 / mcluster.cpp:100-135 -> (qZji, Fz_ji)."""
    K = len(clusters)
    if Xji.shape[0] == 0:
        return np.zeros((0, K)), 0.0
    E = np.zeros(K)
    for t in range(len(weights_t)):
        E = E + qYji[t] * weights_t[t].Elogweight()
    logq = np.empty((Xji.shape[0], K))
    for k in range(K):
        logq[:, k] = E[k] + clusters[k].Eloglike(Xji)
    logZ = logsumexp(logq)
    return np.exp(logq - logZ[:, None]), float(-logZ.sum())


def _resize(lst, n, factory):
    del lst[n:]
    while len(lst) < n:
        lst.append(factory())


def tvbem(X, qZ, qY, weights_j, weights_t, clusters, prior_t, prior_k, maxit=-1, verbose=False, W=None,
          clusters_t=None, trace=None, fixed_iters=-1):
    """scluster.cpp:172-260 / mcluster.cpp:186-287.  qZ and qY are updated in place; returns F."""
    J = len(X)
    K = qZ[0][0].shape[1]
    T = qY[0].shape[1]
    D = X[0][0].shape[1]
    mcm = W is not None
    _resize(weights_j, J, GDirichlet)
    _resize(weights_t, T, (lambda: Dirichlet()) if mcm else (lambda: Dirichlet(prior_t)))
    if mcm:
        _resize(clusters_t, T, lambda: GaussWish(prior_t, W[0].shape[1]))
    _resize(clusters, K, lambda: GaussWish(prior_k, D))
    it = 0
    F = np.finfo(np.float64).max
    while True:
        Fold = F
        Ntk = np.zeros((T, K))
        for j in range(J):
            for i in range(len(X[j])):
                Ntk += np.outer(qY[j][i], qZ[j][i].sum(axis=0))
            weights_j[j].update(qY[j].sum(axis=0))
        for t in range(T):
            if mcm:
                clusters_t[t].clearobs()
                for j in range(J):
                    clusters_t[t].addobs(qY[j][:, t], W[j])
            weights_t[t].update(Ntk[t])
            if mcm:
                clusters_t[t].update()
        for k in range(K):
            clusters[k].clearobs()
            for j in range(J):
                for i in range(len(X[j])):
                    clusters[k].addobs(qZ[j][i][:, k], X[j][i])
            clusters[k].update()
        Fz = Fyz = 0.0
        for j in range(J):
            qY[j], f = vbeY(qZ[j], weights_j[j], weights_t, None, W[j] if mcm else None, clusters_t if mcm else None)
            Fyz += f
        for j in range(J):
            for i in range(len(X[j])):
                qZ[j][i], f = vbeZ(X[j][i], qY[j][i], weights_t, clusters)
                Fz += f
        F = (sum(w.fenergy() for w in weights_j) + sum(w.fenergy() for w in weights_t)
             + (sum(c.fenergy() for c in clusters_t) if mcm else 0.0) + sum(c.fenergy() for c in clusters) + Fyz + Fz)
        if trace is not None:
            trace.append(F)
        if fixed_iters >= 0:
            it += 1
            if it >= fixed_iters:
                return F
            continue
        if (F - Fold) / abs(Fold) > FENGYDEL:
            raise RuntimeError("Free energy increase!")
        if verbose:
            print("-", end="", flush=True)
        if not abs((Fold - F) / Fold) > CONVERGE:
            return F
        it += 1
        if not (it < maxit or maxit < 0):
            return F


def tsplit(X, clusters, prior_t, qY, qZ, tally, F, maxK, verbose, W=None, clusters_t=None, events=None):
    """scluster.cpp:280-428 (split_gr) / mcluster.cpp:309-455 (ssplit)."""
    J = len(X)
    K = len(clusters)
    mcm = W is not None
    if maxK >= 0 and K >= maxK:
        return False
    while len(tally) < K:
        tally.append(0)
    del tally[K:]
    Fk = np.array([c.fenergy() for c in clusters])
    for j in range(J):
        for i in range(len(X[j])):
            for k in range(K):
                if X[j][i].shape[0]:
                    Fk[k] -= float(qZ[j][i][:, k] @ clusters[k].Eloglike(X[j][i]))
    order = std_sort(range(K), lambda i, j: Fk[i] > Fk[j] if tally[i] == tally[j] else tally[i] < tally[j])
    if events is not None:
        events.append(("order", list(order), Fk.tolist()))
    for k in order:
        tally[k] += 1
        if clusters[k].getN() < 4:
            continue
        scount = Mtot = 0
        mapidx = [[None] * len(X[j]) for j in range(J)]
        Xk = [[None] * len(X[j]) for j in range(J)]
        qZref = [[None] * len(X[j]) for j in range(J)]
        for j in range(J):
            for i in range(len(X[j])):
                mapidx[j][i], Xk[j][i] = partobs(X[j][i], qZ[j][i][:, k] > 0.5)
                n = Xk[j][i].shape[0]
                Mtot += n
                splitk = clusters[k].splitobs(Xk[j][i]) if n else np.zeros(0, dtype=bool)
                q = np.zeros((n, 2))
                q[:, 0] = splitk.astype(np.float64)
                q[:, 1] = (~splitk).astype(np.float64)
                qZref[j][i] = q
                scount += int(splitk.sum())
        if scount < 2 or scount > Mtot - 2:
            continue
        wspl, lspl, cspl, ctspl = [], [], [], []
        if mcm:  # mcluster.cpp:414-416: refined with the current qY and W
            qYref = [q.copy() for q in qY]
            tvbem(Xk, qZref, qYref, wspl, lspl, cspl, clusters_t[0].getprior(), clusters[0].getprior(), SPLITITER,
                  W=W, clusters_t=ctspl)
        else:    # scluster.cpp:363, 385-386: one top-level cluster
            qYref = [np.ones((len(X[j]), 1)) for j in range(J)]
            tvbem(Xk, qZref, qYref, wspl, lspl, cspl, prior_t, clusters[0].getprior(), SPLITITER)
        if anyempty(cspl):
            continue
        qZaug = [[auglabels(k, mapidx[j][i], qZref[j][i][:, 1] > 0.5, qZ[j][i]) for i in range(len(X[j]))]
                 for j in range(J)]
        qYaug = [q.copy() for q in qY]
        if mcm:
            Fs = tvbem(X, qZaug, qYaug, wspl, lspl, cspl, clusters_t[0].getprior(), clusters[0].getprior(), 1,
                       W=W, clusters_t=ctspl)
        else:
            Fs = tvbem(X, qZaug, qYaug, wspl, lspl, cspl, prior_t, clusters[0].getprior(), 1)
        if anyempty(cspl):
            continue
        if verbose:
            print("=", end="", flush=True)
        if events is not None:
            events.append(("candidate", int(k), float(Fs)))
        if Fs < F and abs((F - Fs) / F) > CONVERGE:
            for j in range(J):
                qY[j] = qYaug[j]
                qZ[j] = qZaug[j]
            tally[k] = 0
            return True
    return False


def prune_clusters_t(qY, weights_t, clusters_t=None, verbose=False):
    """scluster.cpp:437-481 / mcluster.cpp:465-511: drop top-level clusters with fewer than one observation."""
    Nt = np.array([w.getNk().sum() for w in weights_t])
    empty = Nt < 1
    if not empty.any():
        return False
    if verbose:
        print("*", end="", flush=True)
    keep = np.flatnonzero(~empty)
    for t in sorted(np.flatnonzero(empty), reverse=True):
        del weights_t[t]
        if clusters_t is not None:
            del clusters_t[t]
    for j in range(len(qY)):
        qY[j] = qY[j][:, keep].copy()
    return True


def tcluster(X, prior_t, prior_k, maxT, maxK, verbose, W=None, qY0=None, rng=None, trace=None, events=None):
    """scluster.cpp:493-570 (scluster) / mcluster.cpp:525-605 (mcluster)
    -> (F, qY, qZ, weights_j, weights_t, clusters_t, clusters)."""
    J = len(X)
    mcm = W is not None
    if mcm:
        if len(W) != J:
            raise ValueError("W and X need to have the same number of groups!")
        for j in range(J):
            if W[j].shape[0] != len(X[j]):
                raise ValueError("W and X need to have the same number of 'docs'!")
    if qY0 is not None:
        qY = [np.array(q, dtype=np.float64) for q in qY0]
    else:
        rng = rng or np.random.default_rng(0)
        qY = [random_qY(len(X[j]), maxT, rng) for j in range(J)]
    qZ = [[np.ones((x.shape[0], 1)) for x in X[j]] for j in range(J)]
    Itot = sum(len(x) for x in X)
    if not mcm and maxT > Itot:
        raise ValueError("maxT must be less than the number of documents ofX!")
    weights_j, weights_t, clusters_t, clusters = [], [], [], []
    issplit = emptyclasses = True
    F = 0.0
    tally = []
    while issplit or emptyclasses:
        rtrace = [] if trace is not None else None
        F = tvbem(X, qZ, qY, weights_j, weights_t, clusters, prior_t, prior_k, -1, verbose, W,
                  clusters_t if mcm else None, rtrace)
        if trace is not None:
            trace.append((len(weights_t), len(clusters), rtrace))
        if verbose:
            print("<", end="", flush=True)
        if not issplit:
            emptyclasses = prune_clusters_t(qY, weights_t, clusters_t if mcm else None, verbose)
        else:
            issplit = tsplit(X, clusters, prior_t, qY, qZ, tally, F, maxK, verbose, W, clusters_t if mcm else None,
                             events)
        if verbose:
            print(">")
    if verbose:
        print("Finished!")
        print("Number of top level clusters = %d, and bottom level clusters = %d" % (len(weights_t), len(clusters)))
        print("Free energy =", F)
    return F, qY, qZ, weights_j, weights_t, clusters_t, clusters


def learnSCM(X, dirprior=PRIORVAL, gausprior=PRIORVAL, maxT=TRUNC, maxK=-1, verbose=False, qY0=None, rng=None,
             trace=None, events=None):
    """src/scluster.cpp:578-605 -> (F, qY, qZ, weights_j, weights_t, clusters)."""
    Xl = [[np.asarray(x, dtype=np.float64) for x in Xj] for Xj in X]
    if verbose:
        print("Learning SCM...")
    F, qY, qZ, wj, wt, _, cl = tcluster(Xl, dirprior, gausprior, maxT, maxK, verbose, None, qY0, rng, trace, events)
    return F, qY, qZ, wj, wt, cl


def learnMCM(W, X, prior_t=PRIORVAL, prior_k=PRIORVAL, maxT=TRUNC, maxK=-1, verbose=False, qY0=None, rng=None,
             trace=None, events=None):
    """src/mcluster.cpp:613-642 -> (F, qY, qZ, weights_j, weights_t, clusters_t, clusters_k)."""
    Xl = [[np.asarray(x, dtype=np.float64) for x in Xj] for Xj in X]
    Wl = [np.asarray(w, dtype=np.float64) for w in W]
    if verbose:
        print("Learning MCM...")
    return tcluster(Xl, prior_t, prior_k, maxT, maxK, verbose, Wl, qY0, rng, trace, events)


# ---------------------------------------------------------------------------
# fixed-K harness (no reference entry point: vbem is file-static there,
# cluster.cpp:177).  Used by parity tests and bench.py's cpu_baseline check.
# ---------------------------------------------------------------------------

def suffstats(X, qZ):
    """Dense restatement of K addobs calls: (Nk[K], xs[K,D], xxs[K,D,D])."""
    Nk = qZ.sum(axis=0)
    xs = qZ.T @ X
    xxs = np.einsum("nk,ni,nj->kij", qZ, X, X, optimize=True)
    return Nk, xs, xxs


def vbem_fixed(X, qZ0, wfactory, clusterprior, iters, sparse=False, cfactory=None):
    """``iters`` VBEM iterations from the given qZ with the convergence test
    disabled (same body as vbem, cluster.cpp:198-234).  Returns
    (F trace, Fz trace, qZ list, weights, clusters)."""
    J = len(X)
    K = qZ0[0].shape[1]
    D = X[0].shape[1]
    qZ = [q.copy() for q in qZ0]
    weights = [wfactory() for _ in range(J)]
    clusters = [(cfactory or GaussWish)(clusterprior, D) for _ in range(K)]
    Ftrace, Fztrace = [], []
    for _ in range(iters):
        for c in clusters:
            c.clearobs()
        for j in range(J):
            weights[j].update(updateSS(X[j], qZ[j], clusters, sparse))
        for c in clusters:
            c.update()
        Fz = 0.0
        for j in range(J):
            qZ[j], fz = vbexpectation(X[j], weights[j], clusters, sparse)
            Fz += fz
        Fztrace.append(Fz)
        Ftrace.append(fenergy(weights, clusters, Fz))
    return Ftrace, Fztrace, qZ, weights, clusters
