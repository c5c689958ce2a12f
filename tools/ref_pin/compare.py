#!/usr/bin/env python3
"""Does the oracle agree with the REAL libcluster?  Compares the file tools/ref_pin/ref_dump.cpp wrote on a machine with
Eigen + Boost against tests/golden/xcat_traces.json (generated from oracle/lc_oracle.py by tests/golden/make_golden.py):
F, the number of clusters, the cluster counts, means and covariances, E[log weight] and every responsibility, learner by
learner, at the tolerance BASELINE.json's north star states for F and qZ (1e-5 relative) and at 1e-9 -- the oracle is a
restatement in IEEE doubles of the same operations, so agreement far below 1e-5 is expected, and anything above 1e-9 is
worth a look.  Clusters are matched by their order (both sides keep the order of creation and the reference's final sort).
Usage: tools/ref_pin/compare.py ref_xcat.json      (exit status 0: the oracle is PINNED on these learners)"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
ref = json.loads(Path(sys.argv[1]).read_text())
ora = json.loads((ROOT / "tests" / "golden" / "xcat_traces.json").read_text())
# the separable families' learners on the same data (round 6): their oracle records live in family_traces.json
fam = json.loads((ROOT / "tests" / "golden" / "family_traces.json").read_text())
for _name in ("learnDGMM", "learnDGMC", "learnBEMM", "learnEGMC"):
    ora.setdefault(_name, fam[_name])
worst, bad = 0.0, []
for name, r in ref.items():
    o = ora.get(name)
    if o is None or "F" not in o:
        print(f"{name}: no oracle record to compare with")
        continue
    line = [f"{name}: K ref {r['K']} oracle {o['K']}"]
    if r["K"] != o["K"]:
        bad.append(name + " (cluster count)")
        print(" ".join(line), "MISMATCH")
        continue
    dF = abs(r["F"] - o["F"]) / max(1.0, abs(o["F"]))
    line.append(f"rel dF {dF:.2e}")
    errs = {"F": dF}
    for key in ("N", "means", "covs", "rates", "Elogweight", "qZ"):
        if key not in o:
            continue
        if key in ("Elogweight", "qZ"):  # per group (the single-matrix learners have one)
            e = max(float(np.max(np.abs(np.asarray(a, float) - np.asarray(b, float)))) for a, b in zip(r[key], o[key]))
        else:
            a, b = np.asarray(r[key], float), np.asarray(o[key], float)
            e = float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
        errs[key] = e
        line.append(f"{key} {e:.2e}")
    w = max(errs.values())
    worst = max(worst, w)
    if w > 1e-5:
        bad.append(name)
    print(" ".join(line), "ok" if w <= 1e-9 else ("within 1e-5" if w <= 1e-5 else "OUTSIDE 1e-5"))
print(f"worst deviation {worst:.2e}:", "oracle PINNED on these learners" if not bad else f"NOT pinned: {bad}")
sys.exit(1 if bad else 0)
