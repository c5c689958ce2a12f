import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def estep_cases():
    return json.loads((GOLDEN / "estep_cases.json").read_text())["cases"]


@pytest.fixture(scope="session")
def xcat():
    d = json.loads((GOLDEN / "xcat.json").read_text())
    X = [np.array(g) for g in d["X"]]
    return {"X": X, "Xcat": np.vstack(X), "O": [np.array(g) for g in d["O"]]}


@pytest.fixture(scope="session")
def xcat_traces():
    return json.loads((GOLDEN / "xcat_traces.json").read_text())


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library; built on demand here (CPU cross-compile) so the
    not-gpu suite can check loading/symbols."""
    from libcluster_amd import build, capi

    if not capi.LIB_PATH.exists():
        build.build()
    return capi.lib()
