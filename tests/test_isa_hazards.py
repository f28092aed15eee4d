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
    assert n == 1 and len(bad) == 1 and "no wait state" in bad[0]
    bad, _, _ = t.check({name: [mfma, "s_waitcnt lgkmcnt(0)", swap]})  # a scalar wait is not an MFMA wait state
    assert len(bad) == 1
    for sep in ("s_nop 7", "v_mul_f32_e32 v0, v0, v12"):
        bad, _, _ = t.check({name: [mfma, sep, swap]})
        assert bad == []
    bad, _, _ = t.check({name: [mfma, "s_cbranch_scc1 65535", swap]})  # another straight-line run
    assert bad == []
    bad, _, _ = t.check({"some_other_kernel": ["v_mfma_f32_16x16x16_bf16 v[0:3], v[4:5], v[6:7], v[0:3]"]})
    assert len(bad) == 1 and "K = 16" in bad[0]


def test_product_library_is_clean():
    from vit4hep_amd.build import build

    t = _tool()
    funcs = t.disassemble(build(verbose=False))
    problems, n_guarded, n_swaps = t.check(funcs)
    assert n_guarded >= 1 and n_swaps >= 8, (n_guarded, n_swaps)  # the guarded kernels exist: the check is not vacuous
    assert problems == [], problems
