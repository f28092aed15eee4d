"""CPU: the build-time ISA guard of tools/check_isa_hazards.py - the product library holds no K = 16 MFMA, and no inline-assembly lane swap of the
attention forward sits directly behind an MFMA; plus the checker itself on hand-made instruction lists (it must flag what it is there to flag)."""

import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_isa_hazards", os.path.join(ROOT, "tools", "check_isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_flags_the_two_patterns():
    t = _tool()
    name = "_ZN9v4h_dense21attn_fwd_dense_kernelILi9ELi3ELi2EEEvPKDF16bPS1_Pfiiif"
    mfma = "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]"
    swap = "v_permlane16_swap_b32_e32 v0, v1"
    bad, _, n = t.check({name: [mfma, swap]})
    assert n == 1 and len(bad) == 1 and "wait state" in bad[0]
    bad, _, _ = t.check({name: [mfma, "s_waitcnt lgkmcnt(0)", swap]})  # a scalar wait is one wait state, not eleven
    assert len(bad) == 1
    # a token nop or an UNRELATED vector instruction is not enough for an 8-pass MFMA (11 wait states)
    for sep in (["s_nop 0"], ["s_nop 7"], ["v_mul_f32_e32 v20, v21, v22"], ["s_nop 7", "v_add_f32_e32 v30, v31, v32"]):
        bad, _, _ = t.check({name: [mfma, *sep, swap]})
        assert len(bad) == 1, sep
    # enough wait states by count: s_nop 7 + s_nop 2 = 11; or a compiler-visible vector instruction that READS the result (the compiler pads in front of it)
    for sep in (["s_nop 7", "s_nop 2"], ["v_mul_f32_e32 v0, v0, v12", "v_mul_f32_e32 v1, v1, v12"], ["v_pk_mul_f32 v[0:1], v[0:1], v[12:13]"]):
        bad, _, _ = t.check({name: [mfma, *sep, swap]})
        assert bad == [], sep
    # a dependent instruction that covers only ONE of the two registers the swap reads does not cover the other
    bad, _, _ = t.check({name: [mfma, "v_mul_f32_e32 v0, v0, v12", swap]})
    assert len(bad) == 1
    # an MFMA whose result the swap does not read is no hazard; another straight-line run is none either
    bad, _, _ = t.check({name: ["v_mfma_f32_16x16x32_bf16 v[40:43], v[4:7], v[8:11], v[40:43]", swap]})
    assert bad == []
    bad, _, _ = t.check({name: [mfma, "s_cbranch_scc1 65535", swap]})
    assert bad == []
    bad, _, _ = t.check({"some_other_kernel": ["v_mfma_f32_16x16x16_bf16 v[0:3], v[4:5], v[6:7], v[0:3]"]})
    assert len(bad) == 1 and "K = 16" in bad[0]
    # a branch TARGET between the MFMA and the swap ends the run (addresses as a real disassembly provides them)
    t.ADDRS["tgt_" + name] = {0: 0x100, 1: 0x108, 2: 0x10C, 3: 0x110}
    body = ["s_cbranch_scc1 2", mfma, "s_nop 0", swap]  # the branch at 0x100 jumps to 0x100 + 4 + 8 = 0x10C = the s_nop: the run of the swap starts there
    bad, _, _ = t.check({"tgt_" + name: body})
    assert bad == []


def test_product_library_is_clean():
    from vit4hep_amd.build import build

    t = _tool()
    funcs = t.disassemble(build(verbose=False))
    problems, n_guarded, n_swaps = t.check(funcs)
    assert n_guarded >= 1 and n_swaps >= 100 and any(t.MUST_HAVE in f for f in funcs), (n_guarded, n_swaps)  # every kernel's swaps: the check is not vacuous
    assert problems == [], problems
