"""Build the C-ABI shared library (HIP kernels + host C++) in-tree.

    python -m libcluster_amd.build            # incremental
    python -m libcluster_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  Output:
libcluster_amd/lib/libcluster_hip.so (git-ignored; travels with gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "lib" / "obj"
LIB = PKG / "lib" / "libcluster_hip.so"
# the same library with the fault-injection hooks of the tests compiled in (-DLC_TEST_HOOKS: LC_TEST_CACHE_NO_ROOM,
# LC_TEST_JOURNAL_FAIL_RANK in lc_ctx.cpp); tests load it through LC_LIB_PATH, the shipped library has no such switches
LIB_TESTHOOKS = PKG / "lib" / "libcluster_hip_testhooks.so"
HOOKED_SOURCES = ["lc_ctx.cpp"]
ARCH = "gfx950"

# Device-side scheduling: the AMDGPU register-pressure trackers (and no "unclustered high-RP" re-scheduling stage)
# give the 72-MFMA step of suffstat_kernel a better schedule at its 256-VGPR budget: 25.0 -> 24.2 ms at the
# north-star shape, other kernels unchanged (measured; max-ilp / iterative-ilp / max-memory-clause are slower).
DEVICE_FLAGS = ["-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1"]

SOURCES = ["lc_kernels_estep.hip", "lc_kernels_suffstat.hip", "lc_kernels_diag.hip", "lc_kernels_aux.hip", "lc_kernels_fused.hip",
           "lc_ctx.cpp",
           "lc_comm.cpp", "lc_engine.cpp", "lc_topic.cpp", "lc_capi.cpp"]
HEADERS = ["lc_kernels.h", "lc_device.hpp", "lc_ctx.hpp", "lc_comm.hpp", "lc_engine.hpp", "lc_topic.hpp", "lc_host.hpp",
           "../../include/libcluster_hip.h"]
HASH_STAMP = OBJ / "source_hash.txt"
LAST_BUILD = {"compiled": [], "linked": False}  # what the last build() call did (reported by __graft_entry__.build)


def source_hash() -> str:
    """sha256 over every file the library is built from (csrc/* and the C header), in name order.  Compiled into the
    library (lc_source_hash) so that a stale binary is noticed at load time (capi.lib)."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir() if p.suffix in (".hip", ".cpp", ".hpp", ".h")) + [PKG.parent / "include" / "libcluster_hip.h"]
    for p in files:
        h.update(p.name.encode())
        h.update(b"\0")
        h.update(p.read_bytes())
    return h.hexdigest()[:32]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [CSRC / h for h in HEADERS]
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", f"-I{PKG.parent / 'include'}"]
    common += os.environ.get("LC_EXTRA_CXXFLAGS", "").split()  # kernel tuning experiments (-DLC_SS_UNROLL=2 ...)
    objs, jobs = [], []
    shash = source_hash()
    stale_hash = not HASH_STAMP.exists() or HASH_STAMP.read_text().strip() != shash
    LAST_BUILD["compiled"], LAST_BUILD["linked"] = [], False
    for src in SOURCES:
        s = CSRC / src
        o = OBJ / (Path(src).stem + ".o")
        objs.append(o)
        if force or _newer(o, [s, *hdrs]) or (src == "lc_capi.cpp" and stale_hash):
            cmd = [hipcc, f"--offload-arch={ARCH}", *common, "-c", str(s), "-o", str(o)]
            if src == "lc_capi.cpp":
                cmd.insert(-4, f'-DLC_SOURCE_HASH="{shash}"')
            LAST_BUILD["compiled"].append(src)
            if src.endswith(".hip"):
                cmd[2:2] = DEVICE_FLAGS
            jobs.append(cmd)
    hooked_objs = list(objs)
    for src in HOOKED_SOURCES:
        s = CSRC / src
        o = OBJ / (Path(src).stem + "_testhooks.o")
        hooked_objs[SOURCES.index(src)] = o
        if force or _newer(o, [s, *hdrs]):
            jobs.append([hipcc, f"--offload-arch={ARCH}", *common, "-DLC_TEST_HOOKS", "-c", str(s), "-o", str(o)])
            LAST_BUILD["compiled"].append(src + " (test hooks)")
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, jobs))
    if jobs:
        HASH_STAMP.write_text(shash)
    if force or _newer(LIB, objs):
        # librccl is bound lazily by lc_comm.cpp (dlopen): it is not a link-time dependency
        cmd = [hipcc, "-shared", "-o", str(LIB), *map(str, objs), "-lpthread", "-ldl", "-lrt"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        LAST_BUILD["linked"] = True
    if force or _newer(LIB_TESTHOOKS, hooked_objs):
        subprocess.run([hipcc, "-shared", "-o", str(LIB_TESTHOOKS), *map(str, hooked_objs), "-lpthread", "-ldl", "-lrt"], check=True)
        LAST_BUILD["linked"] = True
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
