"""Build-time guard for two gfx950 hazards found in round 3 (docs/history_r01-r04.md section 5, "What round 3 found" 5a; csrc/v4h_attention_dense.h:12-14,235-236):

  1. `v_mfma_f32_16x16x16_*` accumulating straight onto the result of a 16x16x32 MFMA came out wrong in two of four registers whenever the pair was
     scheduled back to back.  The product library uses no K = 16 MFMA at all: the check is that none appears in its device code.
  2. An inline-assembly instruction that reads an MFMA result gets none of the wait states the compiler inserts for instructions it can see.  The
     instruction-lean attention forward passes its score tiles to `v_permlane16_swap` / `v_permlane32_swap` (inline assembly).  A vector-ALU read of the
     result of an N-pass XDL MFMA needs N + 3 wait states behind it (LLVM's hazard recognizer for gfx940 / gfx950; the 16x16x32 bf16 MFMA has 8 passes:
     11).  For every swap, every MFMA within that distance above it in the same straight-line run whose destination overlaps a register the swap reads must be
     separated from it by EITHER a compiler-visible vector-ALU instruction that itself reads the MFMA's destination (the compiler pads the wait states in
     front of THAT instruction, the swap behind it inherits them - the visible multiply the scores pass through) OR by instructions worth at least the
     required wait states (`s_nop N` = N + 1, any other instruction 1).  A run ends at a branch and at a branch TARGET (another predecessor may jump in).

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
GUARDED_KERNELS = ("",)  # every kernel of the library (round 5: the contraction kernels swap accumulator tiles in their epilogues too); the dense attention forward must be among them
MUST_HAVE = "attn_fwd_dense_kernel"
ADDR = re.compile(r"//\s*([0-9A-Fa-f]{8,16}):")
ADDRS = {}  # symbol -> {instruction index: byte address} (filled by disassemble(); hand-made lists in the tests have none)
BRANCH = ("s_cbranch", "s_branch", "s_endpgm", "s_setpc")
REQUIRED_WAIT_STATES = 11  # 8-pass XDL MFMA (16x16x32 bf16) -> vector-ALU read of its result: passes + 3


def regs(operand):
    """Register numbers of one operand: `v12` -> {12}, `v[4:7]` -> {4 .. 7}; AGPRs are a file of their own (offset 1000); anything else: empty."""
    m = re.fullmatch(r"([va])(\d+)", operand)
    if m:
        return {int(m.group(2)) + (1000 if m.group(1) == "a" else 0)}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", operand)
    if m:
        off = 1000 if m.group(1) == "a" else 0
        return set(range(int(m.group(2)) + off, int(m.group(3)) + 1 + off))
    return set()


def operands(ins):
    parts = ins.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def branch_targets(name, body):
    """Instruction indices of `body` that some branch of the function jumps to (needs the byte addresses of a real disassembly)."""
    at = ADDRS.get(name)
    if not at:
        return set()
    by_addr = {a: i for i, a in at.items()}
    out = set()
    for i, ins in enumerate(body):
        if ins.startswith(("s_cbranch", "s_branch")) and i in at:
            ops = operands(ins)
            try:
                off = int(ops[-1], 0)
            except (ValueError, IndexError):
                continue
            if off >= 32768:
                off -= 65536  # simm16, in dwords, relative to the next instruction
            tgt = at[i] + 4 + 4 * off
            if tgt in by_addr:
                out.add(by_addr[tgt])
    return out


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
                addr = ADDR.search(t)
                t = t.split("//")[0].strip()
                if t:
                    funcs[cur].append(t)
                    if addr:
                        ADDRS.setdefault(cur, {})[len(funcs[cur]) - 1] = int(addr.group(1), 16)
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def wait_states(ins):
    """Wait states an instruction contributes between an MFMA and a later reader of its result."""
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


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
        targets = branch_targets(name, body)
        for k, ins in enumerate(body):
            if not SWAP.search(ins):
                continue
            n_swaps += 1
            read = set().union(*[regs(o) for o in operands(ins)]) if operands(ins) else set()
            ws = 0            # wait states between the instruction under the cursor and the swap
            covered = set()   # registers some compiler-visible vector-ALU instruction in between has read (the compiler padded in front of it)
            j = k - 1
            while j >= 0 and ws < REQUIRED_WAIT_STATES:
                cur = body[j]
                if cur.startswith(BRANCH):
                    break  # start of the straight-line run
                if MFMA.search(cur):
                    ops = operands(cur)
                    dest = regs(ops[0]) if ops else set()
                    hit = dest & read if read else dest  # (a swap whose operands could not be parsed: any MFMA counts)
                    if hit and not (hit <= covered):
                        problems.append(f"{name}: `{ins}` (instruction {k}) reads the result of `{cur}` {ws} wait state(s) behind it with no dependent vector-ALU "
                                        f"instruction in between (needs {REQUIRED_WAIT_STATES}): no wait state the hardware would honour")
                        break
                elif cur.startswith("v_") and not SWAP.search(cur):
                    for o in operands(cur)[1:]:
                        covered |= regs(o)
                ws += wait_states(cur)
                if j in targets:
                    break  # another predecessor may jump in here: the run ends (what precedes on that path is not visible from this scan)
                j -= 1
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
    print(f"{os.path.basename(lib)}: {len(funcs)} functions, {n_mfma} MFMA instructions, {n_guarded} functions checked, {n_swaps} lane swaps")
    if n_guarded == 0 or n_swaps == 0 or not any(MUST_HAVE in f for f in funcs):
        print("check_isa_hazards: the guarded kernels were not found - the check would be vacuous", file=sys.stderr)
        return 2
    for p in problems:
        print("HAZARD: " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
