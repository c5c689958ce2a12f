#!/usr/bin/env python3
"""tests/golden/xcat.json (the reference's test data, test/testdata.h) as the plain text tools/ref_pin/ref_dump.cpp reads:
J, then per group `N_j D` and its rows.  Usage: tools/ref_pin/export_inputs.py [xcat_inputs.txt]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
X = json.loads((ROOT / "tests" / "golden" / "xcat.json").read_text())["X"]
out = Path(sys.argv[1] if len(sys.argv) > 1 else "xcat_inputs.txt")
with out.open("w") as f:
    f.write(f"{len(X)}\n")
    for g in X:
        f.write(f"{len(g)} {len(g[0])}\n")
        for row in g:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
print("wrote", out, "groups", len(X), "rows", sum(len(g) for g in X))
