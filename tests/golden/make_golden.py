"""Generate golden vectors for the E-step / suff-stat path from the numpy
oracle (oracle/lc_oracle.py) and cross-check them against scikit-learn.

Run in the build container:   python tests/golden/make_golden.py
Outputs (committed): tests/golden/estep_cases.json, tests/golden/xcat_traces.json,
tests/golden/family_traces.json, tests/golden/topic_traces.json, tests/golden/scott25_traces.json

The reference itself cannot be built or imported here (no Eigen/Boost), so
these vectors are restatement-derived ("parity unpinned", see oracle header);
scikit-learn's BayesianGaussianMixture E-step pins Eloglike + Dirichlet
weights independently (asserted below before anything is written).
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import lc_oracle as o  # noqa: E402

HERE = Path(__file__).parent


def synth(rng, N, D, K, spread=4.0):
    mus = rng.normal(0, spread, (K, D))
    z = rng.integers(0, K, N)
    X = np.empty((N, D))
    for k in range(K):
        B = rng.normal(size=(D, D))
        L = np.linalg.cholesky(B @ B.T / D + 0.5 * np.eye(D))
        idx = np.flatnonzero(z == k)
        X[idx] = mus[k] + rng.normal(size=(idx.size, D)) @ L.T
    return X, z


def soft_labels(z, K, rng, hard=0.9):
    N = z.size
    if K == 1:
        return np.ones((N, 1))
    q = np.full((N, K), (1 - hard) / (K - 1))
    q[np.arange(N), z] = hard
    q *= rng.uniform(0.8, 1.2, q.shape)
    return q / q.sum(axis=1, keepdims=True)


def sklearn_check(X, q0, prior):
    """Eloglike + Dirichlet Elogweight vs scikit-learn (Bishop 10.2)."""
    from sklearn.mixture import BayesianGaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky

    N, D = X.shape
    K = q0.shape[1]
    w = o.Dirichlet()
    cl = [o.GaussWish(prior, D) for _ in range(K)]
    w.update(o.updateSS(X, q0, cl))
    for c in cl:
        c.update()
    ours = np.stack([c.Eloglike(X) for c in cl], axis=1)
    bgm = BayesianGaussianMixture(n_components=K, covariance_type="full",
                                  weight_concentration_prior_type="dirichlet_distribution")
    bgm.mean_precision_ = np.array([c.beta for c in cl])
    bgm.means_ = np.stack([c.m for c in cl])
    bgm.degrees_of_freedom_ = np.array([c.nu for c in cl])
    # sklearn stores covariances_ = iW / nu and precisions_cholesky_ of that
    bgm.covariances_ = np.stack([c.iW / c.nu for c in cl])
    bgm.precisions_cholesky_ = _compute_precision_cholesky(bgm.covariances_, "full")
    bgm.weight_concentration_ = w.alpha
    theirs = bgm._estimate_log_prob(X)
    # sklearn's log N includes -D/2 log(2 pi) and +D/2 log 2 from the Wishart
    # expectation; libcluster's Eloglike has -D/2 log(pi): identical.
    err = np.abs(ours - theirs).max()
    errw = np.abs(w.Elogweight() - bgm._estimate_log_weights()).max()
    return err, errw


def case(name, rng, J, Ns, D, K, wname, prior=1.0, iters=3, sparse=False, kill=None):
    X, q0 = [], []
    for j in range(J):
        x, z = synth(rng, Ns[j], D, K)
        q = soft_labels(z, K, rng)
        if kill is not None and j == kill[0]:
            # make cluster kill[1] (almost) absent from group j -> sparse path
            q[:, kill[1]] = 1e-6
            q /= q.sum(axis=1, keepdims=True)
        X.append(x)
        q0.append(q)
    wf = {"Dirichlet": o.Dirichlet, "StickBreak": o.StickBreak, "GDirichlet": o.GDirichlet}[wname]
    # first-iteration intermediates
    weights = [wf() for _ in range(J)]
    cl = [o.GaussWish(prior, D) for _ in range(K)]
    Njk = []
    for j in range(J):
        n = o.updateSS(X[j], q0[j], cl, sparse)
        weights[j].update(n)
        Njk.append(n.tolist())
    stats = {"Nk": [c.N_s for c in cl], "xs": [c.x_s.tolist() for c in cl],
             "xxs": [c.xx_s.tolist() for c in cl]}
    for c in cl:
        c.update()
    post = {"nu": [c.nu for c in cl], "beta": [c.beta for c in cl],
            "m": [c.m.tolist() for c in cl], "iW": [c.iW.tolist() for c in cl],
            "logdW": [c.logdW for c in cl], "Fc": [c.fenergy() for c in cl]}
    elogpi = [w.Elogweight().tolist() for w in weights]
    Fw = [w.fenergy() for w in weights]
    ell = [np.stack([c.Eloglike(X[j]) for c in cl], axis=1).tolist() for j in range(J)]
    q1, Fz1 = [], 0.0
    for j in range(J):
        q, fz = o.vbexpectation(X[j], weights[j], cl, sparse)
        q1.append(q.tolist())
        Fz1 += fz
    Ftr, Fztr, qT, _, clT = o.vbem_fixed(X, q0, wf, prior, iters, sparse)
    assert abs(Fztr[0] - Fz1) < 1e-9 * max(1, abs(Fz1))
    if wname == "Dirichlet" and J == 1 and not sparse:
        e, ew = sklearn_check(X[0], q0[0], prior)
        assert e < 1e-9 and ew < 1e-12, (name, e, ew)
        print(f"  {name}: sklearn |dEloglike|={e:.2e} |dElogw|={ew:.2e}")
    return {
        "name": name, "J": J, "D": D, "K": K, "weights": wname, "prior": prior,
        "iters": iters, "sparse": sparse,
        "X": [x.tolist() for x in X], "q0": [q.tolist() for q in q0],
        "Njk": Njk, "stats": stats, "post": post, "Elogpi": elogpi, "Fw": Fw,
        "Eloglike": ell, "q1": q1, "Fz1": Fz1,
        "Ftrace": Ftr, "Fztrace": Fztr, "qT": [q.tolist() for q in qT],
        "NkT": [c.getN() for c in clT],
    }


def main():
    rng = np.random.default_rng(20261001)
    cases = [
        case("d2_k3_dir", rng, 1, [60], 2, 3, "Dirichlet"),
        case("d3_k2_sb", rng, 1, [50], 3, 2, "StickBreak"),
        case("d16_k8_dir", rng, 1, [128], 16, 8, "Dirichlet"),
        case("d23_k5_gdir_j3", rng, 3, [40, 33, 27], 23, 5, "GDirichlet"),
        case("d64_k4_sb", rng, 1, [96], 64, 4, "StickBreak", iters=2),
        case("d5_k1_dir_ragged", rng, 1, [17], 5, 1, "Dirichlet"),
        case("d4_k4_gdir_sparse", rng, 2, [45, 38], 4, 4, "GDirichlet", sparse=True, kill=(1, 2)),
        case("d2_k2_dir_prior", rng, 1, [31], 2, 2, "Dirichlet", prior=0.37),
    ]
    (HERE / "estep_cases.json").write_text(json.dumps({"cases": cases}))

    d = json.loads((HERE / "xcat.json").read_text())
    X = [np.array(g) for g in d["X"]]
    Xcat = np.vstack(X)
    out = {}
    for name, fn, arg in (("learnBGMM", o.learnBGMM, Xcat), ("learnVDP", o.learnVDP, Xcat),
                          ("learnGMC", o.learnGMC, X), ("learnSGMC", o.learnSGMC, X)):
        tr, ev = [], []
        F, qZ, w, cl = fn(arg, trace=tr, events=ev)
        wl = w if isinstance(w, list) else [w]
        qs = qZ if isinstance(qZ, list) else [qZ]
        out[name] = {
            "F": F, "K": len(cl), "rounds": [[k, t] for k, t in tr], "events": ev,
            "N": [c.getN() for c in cl], "means": [c.getmean().tolist() for c in cl],
            "covs": [c.getcov().tolist() for c in cl],
            "Elogweight": [x.Elogweight().tolist() for x in wl],
            "qZ": [q.tolist() for q in qs],
        }
        print(name, "F =", F, "K =", len(cl))
    F, _, _, cl = o.learnBGMM(Xcat, maxclusters=1)
    out["learnBGMM_max1"] = {"F": F, "K": len(cl)}
    # sparse GMC on this toy data hits the reference's own guard
    # (cluster.cpp:229-230) in the K=4 round: recorded as an expected error.
    tr = []
    try:
        o.learnGMC(X, sparse=True, trace=tr)
        out["learnGMC_sparse"] = {"throws": None}
    except RuntimeError as e:
        out["learnGMC_sparse"] = {"throws": str(e), "rounds": [[k, t] for k, t in tr]}
    F, _, _, cl = o.learnVDP(Xcat, weights=o.StickBreak(2.5))
    out["learnVDP_conc2.5"] = {"F": F, "K": len(cl)}
    (HERE / "xcat_traces.json").write_text(json.dumps(out))

    # diagonal-Gaussian and exponential families (NormGamma / ExpGamma): the toy data, the toy data folded to
    # [0.1, inf) for the exponential learners, and a 3-component exponential mixture in 4 dimensions
    fam = {}
    Xpos = [np.abs(g) + 0.1 for g in X]
    rates = np.array([[0.2, 5.0, 1.0, 0.5], [4.0, 0.25, 2.0, 6.0], [1.0, 1.0, 8.0, 0.1]])
    zz = rng.integers(0, 3, size=360)
    Xexp = rng.exponential(1.0, size=(360, 4)) / rates[zz]
    fam["Xexp"] = Xexp.tolist()
    Xexp_g = [Xexp[:130], Xexp[130:250], Xexp[250:]]
    for name, fn, arg in (("learnDGMM", o.learnDGMM, Xcat), ("learnDGMC", o.learnDGMC, X),
                          ("learnBEMM", o.learnBEMM, np.vstack(Xpos)), ("learnEGMC", o.learnEGMC, Xpos),
                          ("learnBEMM_exp", o.learnBEMM, Xexp), ("learnEGMC_exp", o.learnEGMC, Xexp_g)):
        tr = []
        F, qZ, w, cl = fn(arg, trace=tr)
        wl = w if isinstance(w, list) else [w]
        qs = qZ if isinstance(qZ, list) else [qZ]
        rec = {"F": F, "K": len(cl), "rounds": [[k, t] for k, t in tr], "N": [c.getN() for c in cl],
               "Elogweight": [x.Elogweight().tolist() for x in wl], "qZ": [q.tolist() for q in qs]}
        if hasattr(cl[0], "getrate"):
            rec["rates"] = [c.getrate().tolist() for c in cl]
        else:
            rec["means"] = [c.getmean().tolist() for c in cl]
            rec["covs"] = [c.getcov().tolist() for c in cl]
        fam[name] = rec
        print(name, "F =", F, "K =", len(cl))
    (HERE / "family_traces.json").write_text(json.dumps(fam))

    # two-level models on the reference's own test set-up (test/scluster_test.cpp:44-68: the 12 groups of
    # testdata.h as 2 groups x 6 documents, maxT = 4; test/mcluster_test.cpp:44-70: + the O data, maxT = 10).
    # The reference's start is std::rand(); the start used is stored with the trace.
    top = {}
    Xv = [X[:6], X[6:]]
    Wd = [np.array(g) for g in d["O"]]
    for name, maxT, Wx in (("learnSCM", 4, None), ("learnMCM", 10, Wd)):
        qY0 = [o.random_qY(6, maxT, rng) for _ in range(2)]
        tr = []
        if Wx is None:
            F, qY, qZ, wj, wt, cl = o.learnSCM(Xv, maxT=maxT, qY0=qY0, trace=tr)
            ct = []
        else:
            F, qY, qZ, wj, wt, ct, cl = o.learnMCM(Wx, Xv, maxT=maxT, qY0=qY0, trace=tr)
        top[name] = {
            "qY0": [q.tolist() for q in qY0], "maxT": maxT, "F": F, "T": len(wt), "K": len(cl),
            "rounds": [[t, k, f] for t, k, f in tr], "qY": [q.tolist() for q in qY],
            "qZ": [[q.tolist() for q in qj] for qj in qZ],
            "Elogweight_j": [w.Elogweight().tolist() for w in wj],
            "Elogweight_t": [w.Elogweight().tolist() for w in wt],
            "means_k": [c.getmean().tolist() for c in cl], "covs_k": [c.getcov().tolist() for c in cl],
            "means_t": [c.getmean().tolist() for c in ct], "covs_t": [c.getcov().tolist() for c in ct],
        }
        print(name, "F =", F, "T =", len(wt), "K =", len(cl))
    (HERE / "topic_traces.json").write_text(json.dumps(top))

    # scott25.dat: the 9815 x 23 data file the reference keeps next to its tests (test/scott25.dat; committed
    # here as a data fixture).  Flat learners on all rows, grouped learners on five consecutive blocks, the
    # two-level learner on 2 groups x 5 documents of ~980 rows.
    S = np.loadtxt(HERE / "scott25.dat", skiprows=2)
    cuts = [0, 1500, 3800, 5200, 7900, S.shape[0]]
    Sg = [S[cuts[i]:cuts[i + 1]] for i in range(5)]
    sc = {}
    for name, fn, arg in (("learnBGMM", o.learnBGMM, S), ("learnVDP", o.learnVDP, S), ("learnDGMM", o.learnDGMM, S),
                          ("learnBEMM", o.learnBEMM, S), ("learnGMC", o.learnGMC, Sg), ("learnDGMC", o.learnDGMC, Sg)):
        tr = []
        F, qZ, w, cl = fn(arg, trace=tr)
        wl = w if isinstance(w, list) else [w]
        sc[name] = {"F": F, "K": len(cl), "rounds": [[k, t] for k, t in tr], "N": [c.getN() for c in cl],
                    "Elogweight": [x.Elogweight().tolist() for x in wl]}
        if hasattr(cl[0], "getrate"):
            sc[name]["rates"] = [c.getrate().tolist() for c in cl]
        else:
            sc[name]["means"] = [c.getmean().tolist() for c in cl]
        print("scott25", name, "F =", F, "K =", len(cl))
    docs = np.array_split(S, 10)
    Xd = [docs[:5], docs[5:]]
    qY0 = [o.random_qY(5, 4, rng) for _ in range(2)]
    tr = []
    F, qY, qZ, wj, wt, cl = o.learnSCM(Xd, maxT=4, qY0=qY0, trace=tr)
    sc["learnSCM"] = {"qY0": [q.tolist() for q in qY0], "F": F, "T": len(wt), "K": len(cl),
                      "rounds": [[t, k, f] for t, k, f in tr], "qY": [q.tolist() for q in qY],
                      "means": [c.getmean().tolist() for c in cl]}
    print("scott25 learnSCM F =", F, "T =", len(wt), "K =", len(cl))
    (HERE / "scott25_traces.json").write_text(json.dumps(sc))
    for f in ("estep_cases.json", "xcat_traces.json", "family_traces.json", "topic_traces.json", "scott25_traces.json"):
        print(f, (HERE / f).stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
