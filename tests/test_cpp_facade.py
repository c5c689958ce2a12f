"""The drop-in C++ headers (include/libcluster.h, include/distributions.h):
they must compile with plain g++ against the C-ABI library (CPU check), and
the reference's own test main, re-written with assertions, must pass on the GPU."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "tests" / "cpp" / "_build" / "cluster_test"


def _compile(lib):
    from libcluster_amd import capi

    EXE.parent.mkdir(exist_ok=True)
    libdir = capi.LIB_PATH.parent
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", f"-I{ROOT / 'include'}", str(ROOT / "tests/cpp/cluster_test.cpp"),
           "-o", str(EXE), f"-L{libdir}", "-lcluster_hip", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_headers_compile_and_link_with_gxx(lib):
    _compile(lib)


def test_c_header_is_plain_c(lib, tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "libcluster_hip.h"\nint main(void){ return lc_version() > 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", "-c", str(src), "-o",
                        str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [None, 3])
def test_reference_test_main_passes_on_gpu(lib, xcat, xcat_traces, gpus):
    """gpus = 3: the same main with LIBCLUSTER_GPUS=3 -- every learn*() call with enough groups / rows shards itself over
    three contexts (all on GPU 0 here, host-staged sums) and must still satisfy every assertion, qZ included (the
    facade fetches it through lc_model_get_qz_all_colmajor)."""
    import os

    exe = _compile(lib)
    env = dict(os.environ)
    if gpus:
        env.update({"LIBCLUSTER_GPUS": str(gpus), "LIBCLUSTER_GPUS_SAME_DEVICE": "1"})
    X = xcat["X"]
    lines = [f"{len(X)} {X[0].shape[0]} {X[0].shape[1]}"]
    for g in X:
        lines += [" ".join(repr(float(v)) for v in row) for row in g]
    lines.append(f"{xcat_traces['learnGMC']['F']!r} {xcat_traces['learnBGMM']['F']!r} {xcat_traces['learnVDP']['F']!r}")
    fam = json.loads((ROOT / "tests" / "golden" / "family_traces.json").read_text())
    lines.append(" ".join(repr(fam[k]["F"]) for k in ("learnDGMM", "learnDGMC", "learnBEMM", "learnEGMC")))
    lines.append(str(len(X)))  # the O data of testdata.h (document observations of mcluster_test.cpp)
    for g in xcat["O"]:
        lines += [" ".join(repr(float(v)) for v in row) for row in g]
    r = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "cluster_test OK" in r.stdout
    # the reference's verbose progress glyphs (README.md:355-376, cluster.cpp:232-233, 605-617)
    assert "Learning GMC..." in r.stdout and "<" in r.stdout and ">" in r.stdout and "Finished!" in r.stdout
    assert "Learning SCM..." in r.stdout and "Learning MCM..." in r.stdout
    assert "Number of top level clusters = " in r.stdout and ", and bottom level clusters = " in r.stdout
    # LIBCLUSTER_GPUS=8 inside the main (set with setenv between the calls): eight row blocks / sixteen groups on eight
    # shards equal to the unsharded calls
    assert "Sharding over 8 GPU(s), host-local all-reduce" in r.stdout and "eight shards OK" in r.stdout
