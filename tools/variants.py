#!/usr/bin/env python3
"""Kernel experiments: build named variants of the library (extra -D flags on the device sources) here, then time
them all in ONE GPU call.

    tools/variants.py build NAME [-DFLAG ...]     # -> tools/variants/NAME.so (git-ignored, travels with gpurun)
    tools/variants.py run [--shape N,D,K] [--what estep,suffstat,diag] [NAME ...]   # on the GPU box

`run` starts one child process per variant (LC_LIB_PATH), which synthesises the north-star shape, runs three fixed
VBEM iterations (F printed to 15 digits: a variant that changes results shows here) and reports the average
E-step / statistics kernel times over the timed launches."""
import json
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
VDIR = ROOT / "tools" / "variants"
sys.path.insert(0, str(ROOT))


def build(name, flags):
    from libcluster_amd import build as b

    VDIR.mkdir(exist_ok=True)
    obj = VDIR / (name + "_obj")
    obj.mkdir(exist_ok=True)
    hipcc = b._hipcc()
    common = ["-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}"]
    objs, jobs = [], []
    for src in b.SOURCES:
        if src.endswith(".hip"):
            o = obj / (Path(src).stem + ".o")
            jobs.append([hipcc, f"--offload-arch={b.ARCH}", *b.DEVICE_FLAGS, *common, *flags, "-c", str(b.CSRC / src), "-o", str(o)])
        else:
            o = b.OBJ / (Path(src).stem + ".o")  # host objects of the current default build
        objs.append(o)
    from concurrent.futures import ThreadPoolExecutor

    only = os.environ.get("LC_VARIANT_ONLY")  # e.g. "estep": compile just that device source, take the others from the default build
    if only:
        keep = []
        for cmd in jobs:
            if any(t in cmd[-3] for t in only.split(",")):
                keep.append(cmd)
            else:
                shutil.copy2(b.OBJ / Path(cmd[-1]).name, cmd[-1])
        jobs = keep
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda c: subprocess.run(c, check=True), jobs))
    out = VDIR / (name + ".so")
    subprocess.run([hipcc, "-shared", "-o", str(out), *map(str, objs), "-lpthread", "-ldl", "-lrt"], check=True)
    shutil.rmtree(obj)
    print("built", out)


CHILD = r"""
import sys, json, time
sys.path.insert(0, {root!r})
import numpy as np
import torch
from libcluster_amd import capi
import bench
N, D, K = {N}, {D}, {K}
fam = {fam!r}
mu, L = bench.mixture(D, K, 1004)
ck = {{"gw": capi.C_GAUSSWISH, "ng": capi.C_NORMGAMMA, "eg": capi.C_EXPGAMMA}}[fam]
if fam != "gw":
    rng = np.random.default_rng(1006)
    mu = rng.uniform(20.0, 60.0, (K, D)) if fam == "eg" else rng.normal(0.0, 3.0, (K, D))
    L = np.stack([np.diag(rng.uniform(0.5, 1.5, D)) for _ in range(K)])
with capi.Context(0) as ctx:
    ctx.synth(N, D, K, mu, L, 1004, 0, 0.9)
    F, tr, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters=3, nthreads=8, ckind=ck)
    m.close()
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    F2, tr2, m = ctx.vbem(capi.W_DIRICHLET, fixed_iters={iters}, nthreads=8, ckind=ck)
    dt = time.perf_counter() - t0
    t = ctx.timing_get()
    m.close()
print("RESULT " + json.dumps(dict(F=repr(float(tr[-1])), F2=repr(float(tr2[-1])), iter_ms=dt * 1e3 / {iters},
      estep_ms=t["estep_ms"] / max(1, t["estep_calls"]), suffstat_ms=t["suffstat_ms"] / max(1, t["suffstat_calls"]),
      fused_ms=t["fused_ms"] / max(1, t["fused_calls"]))))
"""


def run(argv):
    shapes, fam, iters, names = [(10_000_000, 64, 32)], "gw", 12, []
    it = iter(argv)
    for a in it:
        if a == "--shape":  # "N,D,K" or several: "N,D,K;N,D,K"
            shapes = [tuple(int(x) for x in sh.split(",")) for sh in next(it).split(";")]
        elif a == "--fam":
            fam = next(it)
        elif a == "--iters":
            iters = int(next(it))
        else:
            names.append(a)
    libs = [("default", ROOT / "libcluster_amd" / "lib" / "libcluster_hip.so")]
    libs += sorted((p.stem, p) for p in VDIR.glob("*.so") if not names or p.stem in names)
    rep = int(os.environ.get("LC_VARIANT_REPEAT", "1"))
    for shape in shapes:
        print(f"shape N,D,K = {shape} family {fam}, {iters} timed iterations")
        for r in range(rep):
            for name, path in libs:
                e = dict(os.environ, LC_LIB_PATH=str(path), LC_ALLOW_STALE_LIB="1")
                code = CHILD.format(root=str(ROOT), N=shape[0], D=shape[1], K=shape[2], fam=fam, iters=iters)
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
                if p.returncode != 0 or not line:
                    print(f"{name:28s} FAILED rc={p.returncode} {p.stderr[-400:]}")
                    continue
                d = json.loads(line[-1][7:])
                print(f"{name:28s} estep {d['estep_ms']:7.3f} ms  suffstat {d['suffstat_ms']:7.3f} ms  fused {d['fused_ms']:6.3f}  "
                      f"iter {d['iter_ms']:7.3f} ms  F {d['F']} {d['F2']}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2:])
