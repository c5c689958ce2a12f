#!/usr/bin/env python3
"""Build-time checks on the generated gfx950 ISA of estep_kernel (tools/check_isa.py [file.s]).

The kernel's cluster loop does three things behind the compiler's back (lc_kernels_estep.hip):
  * it writes M0 in inline asm for `global_load_lds_dwordx4` (hipcc refuses M0 in a clobber list: "reserved register");
  * it loads c_jk with inline-asm `global_load_dwordx2`, invisible to hipcc's s_waitcnt insertion, and waits for them
    in a later asm statement -- correct only while no copy / spill / use of those registers lands in between;
  * its register budget (three waves per SIMD at D = 64) has no room for scratch.
This script compiles the device code to assembly (or reads a given .s) and asserts, for every estep_kernel instance:
  1. private_segment_fixed_size == 0 (no scratch, no spills);
  2. every instruction that names m0 is an `s_mov_b32 m0, ...` directly followed by a `global_load_lds_dwordx4`
     (nothing else in the kernel depends on M0, so writing it unannounced is safe);
  3. between an inline-asm `global_load_dwordx2 vA, vOff, s[..]` and the next `s_waitcnt vmcnt(0)` no instruction reads
     or writes the destination registers vA.
And for every kernel of lc_kernels_estep / _fused / _diag.hip:
  4. no inline-asm statement reads a VGPR that a v_mfma wrote fewer than MFMA_WAIT wait states earlier.  The hazard
     recogniser counts the wait states between an MFMA and a VALU reader only for instructions it emitted itself; an
     asm `v_max_f64` right behind the last link of an MFMA chain read a stale register (NaN rows out of the half-width
     fused instance, round 4).  MFMA results go through compiler-generated instructions (fmax) before any asm sees them.
     The other direction as well (round 5): no inline-asm statement WRITES a VGPR that an MFMA inside that window has as
     its destination or reads as SrcC (an asm output the register allocator placed in a just-freed register).
     The walk is linear over the text of a function and does not follow branches: a hazard across a loop back-edge is
     outside what it sees.
Exit status 0 = all hold.  Run by tests/test_host.py (CPU: hipcc cross-compiles)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


MFMA_WAIT = 19  # the longest MFMA -> VALU-read distance of the CDNA3/4 hazard tables (16-pass); the 4x4x4 f64 needs fewer
# an MFMA reads SrcC in its first pass(es): the write-after-read window of the tables is 11 wait states for the 16-pass
# instructions and 5 for the 4-pass ones (v_mfma_f64_4x4x4 is a 4-pass instruction); 11 covers every instruction these
# kernels use with room to spare
MFMA_WAR_WAIT = 11
ASM_FILES = ("lc_kernels_estep.hip", "lc_kernels_fused.hip", "lc_kernels_diag.hip")


def device_asm(name="lc_kernels_estep.hip", src=None) -> str:
    sys.path.insert(0, str(ROOT))
    from libcluster_amd import build as b

    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = [b._hipcc(), f"--offload-arch={b.ARCH}", *b.DEVICE_FLAGS, "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}",
               f"-I{b.CSRC}", "-S", "--cuda-device-only", str(src or b.CSRC / name), "-o", str(out)]
        subprocess.run(cmd, check=True, capture_output=True)
        return out.read_text()


def check_mfma_into_asm(asm: str):
    """Check 4 over every function of one assembly file -> (problems, number of asm statements looked at)."""
    problems, nasm = [], 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2).splitlines()
        recent = []    # (registers written by an MFMA, wait states since)
        recent_c = []  # (registers an MFMA reads as SrcC, wait states since)
        in_asm = False
        for ln in body:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            op, ops = operands(ln)
            if not op:
                continue
            if in_asm:
                nasm += 1
                # stores have no destination; everything else writes its first operand and reads the rest
                srcs = ops if op.startswith(("global_store", "ds_write", "buffer_store")) else ops[1:]
                read = set().union(*[regs(o) for o in srcs]) if srcs else set()
                wrote = set() if op.startswith(("global_store", "ds_write", "buffer_store", "s_")) or not ops else regs(ops[0])
                for wr, age in recent:
                    if wr & read and age < MFMA_WAIT:
                        problems.append(f"{name}: inline asm `{t}` reads v{sorted(wr & read)} {age} wait states after the MFMA that "
                                        f"wrote them (needs {MFMA_WAIT} or a compiler-generated reader in between)")
                    if wr & wrote and age < MFMA_WAIT:
                        problems.append(f"{name}: inline asm `{t}` writes v{sorted(wr & wrote)} {age} wait states after an MFMA whose "
                                        f"destination they are (no wait states are inserted for asm in this direction either)")
                for rd, age in recent_c:
                    if rd & wrote and age < MFMA_WAR_WAIT:
                        problems.append(f"{name}: inline asm `{t}` writes v{sorted(rd & wrote)} {age} wait states after an MFMA that "
                                        f"reads them as SrcC")
            step = 1
            if op == "s_nop" and ops and ops[0].isdigit():
                step = int(ops[0]) + 1
            recent = [(wr, age + step) for wr, age in recent if age + step < MFMA_WAIT]
            recent_c = [(rd, age + step) for rd, age in recent_c if age + step < MFMA_WAIT]
            if op.startswith("v_mfma") and ops:
                recent.append((regs(ops[0]), 0))
                if len(ops) >= 4 and regs(ops[3]):
                    recent_c.append((regs(ops[3]), 0))
            elif not in_asm and ops:
                # a compiler-generated instruction that overwrites an MFMA result ends that result's hazard window
                w = regs(ops[0])
                recent = [(wr - w, age) for wr, age in recent if wr - w]
    return problems, nasm


def regs(tok: str):
    """VGPR numbers named by one operand token (v12, v[4:5]); empty for anything else."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(line: str):
    body = line.split(";")[0].strip()
    if not body or body.endswith(":") or body.startswith("."):
        return None, []
    parts = body.split(None, 1)
    ops = [t.strip() for t in re.split(r",\s*|\s+", parts[1])] if len(parts) > 1 else []
    return parts[0], ops


def check(asm: str):
    problems, seen = [], 0
    scratch = dict(re.findall(r"\.amdhsa_kernel\s+(\S+)\s+\.amdhsa_group_segment_fixed_size\s+\d+\s+"
                              r"\.amdhsa_private_segment_fixed_size\s+(\d+)", asm))
    for name, sz in scratch.items():
        if "estep_kernel" in name and int(sz) != 0:
            problems.append(f"{name}: {sz} bytes of scratch")
    if not any("estep_kernel" in n for n in scratch):
        problems.append("no .amdhsa_kernel descriptor of an estep_kernel instance found")
    for m in re.finditer(r"^(_ZN3lck12estep_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2).splitlines()
        seen += 1
        ins = [(i, *operands(ln)) for i, ln in enumerate(body)]
        ins = [(i, op, ops) for i, op, ops in ins if op]
        for n, (i, op, ops) in enumerate(ins):
            if any(o == "m0" for o in ops):
                nxt = ins[n + 1][1] if n + 1 < len(ins) else ""
                if not (op == "s_mov_b32" and ops and ops[0] == "m0" and nxt.startswith("global_load_lds_dwordx4")):
                    problems.append(f"{name}: `{body[i].strip()}` names m0 outside the LDS-direct load pair")
        pending = set()  # destination registers of asm loads not yet waited for
        for n, (i, op, ops) in enumerate(ins):
            if op == "s_waitcnt" and any(o.startswith("vmcnt(0)") for o in ops):
                pending = set()
                continue
            touched = set().union(*[regs(o) for o in ops]) if ops else set()
            if pending & touched:
                problems.append(f"{name}: `{body[i].strip()}` touches {sorted(pending & touched)} before the s_waitcnt vmcnt(0) "
                                "that covers their inline-asm load")
            # the c_jk loads: scalar-base form with a VGPR offset (compiler-generated loads of this kernel use either
            # `off` or a 64-bit VGPR address; hipcc's own waitcnt pass covers those)
            if op == "global_load_dwordx2" and len(ops) >= 3 and re.fullmatch(r"v\d+", ops[1]) and ops[2].startswith("s["):
                pending |= regs(ops[0])
    if seen == 0:
        problems.append("no estep_kernel instance found in the assembly")
    return problems, seen


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--mfma-asm":  # check 4 alone, on a given source file (or .s)
        p = Path(sys.argv[2])
        problems, nasm = check_mfma_into_asm(p.read_text() if p.suffix == ".s" else device_asm(src=p))
        for q in problems:
            print("ISA check:", q)
        print(f"checked {nasm} inline-asm statements behind MFMAs: {'FAILED' if problems else 'ok'}")
        sys.exit(1 if problems else 0)
    if len(sys.argv) > 1:
        asms = {"given": Path(sys.argv[1]).read_text()}
    else:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(len(ASM_FILES)) as ex:
            asms = dict(zip(ASM_FILES, ex.map(device_asm, ASM_FILES)))
    first = next(iter(asms.values()))
    problems, seen = check(first)
    nasm = 0
    for name, asm in asms.items():
        p4, n4 = check_mfma_into_asm(asm)
        problems += [f"[{name}] {q}" for q in p4]
        nasm += n4
    for p in problems:
        print("ISA check:", p)
    print(f"checked {seen} estep_kernel instances: {'FAILED' if problems else 'ok'}")
    print(f"checked {nasm} inline-asm statements behind MFMAs in {len(asms)} files: {'FAILED' if problems else 'ok'}")
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
