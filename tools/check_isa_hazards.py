"""Build-time guard for two gfx950 hazards found in round 3 (docs/history_r01-r04.md section 5, "What round 3 found" 5a; csrc/v4h_attention_dense.h:12-14,235-236):

  1. `v_mfma_f32_16x16x16_*` accumulating straight onto the result of a 16x16x32 MFMA came out wrong in two of four registers whenever the pair was
     scheduled back to back.  The product library uses no K = 16 MFMA at all: the check is that none appears in its device code.
  2. An inline-assembly instruction that reads an MFMA result gets none of the wait states the compiler inserts for instructions it can see.  The
     instruction-lean attention forward passes its score tiles to `v_permlane16_swap` / `v_permlane32_swap` (inline assembly): between the last MFMA
     above such a swap and the swap itself there must be wait states the source put there - an `s_nop`, or a vector-ALU instruction (the visible
     multiply the results pass through), never the MFMA immediately.

Works on the shipped library (no GPU needed): the gfx950 code objects are unbundled into a scratch directory with llvm-objdump --offloading and
disassembled.  Exit code 0 = clean, 1 = a hazard pattern was found (printed), 2 = tooling problem.

    python tools/check_isa_hazards.py [path/to/libvit4hep_hip.so]
"""

from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
SWAP = re.compile(r"\bv_permlane(16|32)_swap")
MFMA = re.compile(r"\bv_mfma_")
K16 = re.compile(r"\bv_mfma_f32_16x16x16")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
GUARDED_KERNELS = ("attn_fwd_dense_kernel",)


def disassemble(lib):
    """{symbol: [instruction text, ...]} of every function in the library's gfx950 code objects."""
    tmp = tempfile.mkdtemp(prefix="v4h_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)  # llvm-objdump writes the unbundled code objects next to its input
        r = subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump --offloading failed: {r.stderr[-400:]}")
        cos = sorted(glob.glob(local + ".*gfx950*"))
        if not cos:
            raise RuntimeError("no gfx950 code object found in " + lib)
        funcs = {}
        for co in cos:
            d = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
            if d.returncode != 0:
                raise RuntimeError(f"llvm-objdump -d failed on {co}: {d.stderr[-400:]}")
            cur = None
            for ln in d.stdout.splitlines():
                m = LABEL.match(ln)
                if m:
                    cur = m.group(1)
                    funcs.setdefault(cur, [])
                    continue
                t = ln.strip()
                if cur is None or not t or t.startswith(("//", ";", "Disassembly", "/")):
                    continue
                t = re.sub(r"^[0-9a-f]+:\s*", "", t)  # (address prefix, when present)
                t = t.split("//")[0].strip()
                if t:
                    funcs[cur].append(t)
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def is_wait_state(ins):
    """An instruction that separates an MFMA from an inline-assembly reader: s_nop, or any vector-ALU instruction that is not itself an MFMA / swap."""
    return ins.startswith("s_nop") or (ins.startswith("v_") and not MFMA.search(ins) and not SWAP.search(ins))


def check(funcs):
    problems = []
    n_swaps = n_guarded = 0
    for name, body in funcs.items():
        for k, ins in enumerate(body):
            if K16.search(ins):
                problems.append(f"{name}: K = 16 MFMA in the product library: `{ins}`")
        if not any(g in name for g in GUARDED_KERNELS):
            continue
        n_guarded += 1
        for k, ins in enumerate(body):
            if not SWAP.search(ins):
                continue
            n_swaps += 1
            j = k - 1
            seen_wait = False
            while j >= 0 and not MFMA.search(body[j]):
                if body[j].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    j = -1  # start of the straight-line run: no MFMA above the swap in it
                    break
                seen_wait = seen_wait or is_wait_state(body[j])
                j -= 1
            if j >= 0 and not seen_wait:
                problems.append(f"{name}: `{ins}` (instruction {k}) directly behind `{body[j]}` with no wait state in between")
    return problems, n_guarded, n_swaps


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vit4hep_amd", "libvit4hep_hip.so")
    try:
        funcs = disassemble(lib)
    except (OSError, RuntimeError) as e:
        print(f"check_isa_hazards: {e}", file=sys.stderr)
        return 2
    problems, n_guarded, n_swaps = check(funcs)
    n_mfma = sum(1 for b in funcs.values() for i in b if MFMA.search(i))
    print(f"{os.path.basename(lib)}: {len(funcs)} functions, {n_mfma} MFMA instructions, {n_guarded} guarded attention kernels with {n_swaps} lane swaps")
    if n_guarded == 0 or n_swaps == 0:
        print("check_isa_hazards: the guarded kernels were not found - the check would be vacuous", file=sys.stderr)
        return 2
    for p in problems:
        print("HAZARD: " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
