"""-m gpu: per-kernel parity of libvit4hep_hip.so against fp64 PyTorch math / the CPU oracle.

Tolerances: f32 mode = exact-f32 MFMA, only summation order differs  -> 2e-5 relative;
            bf16 mode = bf16 operands, f32 accumulate              -> exact on small integers, 2e-2 on random data.
"""

import math

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from tests import hiputil as U

pytestmark = pytest.mark.gpu
MODES = ["f32", "bf16"]


def _ints(shape, lo, hi, dtype, gen):
    return torch.randint(lo, hi + 1, shape, generator=gen).to(dtype).to(U.DEV)


# ----------------------------------------------------------------------------------------------- contractions
LAYOUTS = {"fwd": (0, 0), "dgrad": (0, 1), "wgrad": (1, 1)}
SHAPES = [(270, 480, 480), (128, 160, 32), (135, 1440, 480), (17, 96, 64), (300, 64, 1920), (16, 480, 256)]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("I,J,K", SHAPES)
def test_gemm_exact_integers(mode, layout, I, J, K):
    """Small-integer operands are exact in bf16 and f32: any fragment-layout / indexing slip shows as a wrong integer.
    Asymmetric operands (P != Q^T, values depend on position) so that a transposed output cannot pass."""
    pks, qks = LAYOUTS[layout]
    if layout == "wgrad":
        I, K = (I + 7) // 8 * 8, K  # K-strided P needs I in whole 16-byte chunks
        I = max(I, 8)
    g = torch.Generator().manual_seed(I * 7 + J * 3 + K)
    dt = U.tdtype(mode)
    Pm = _ints((I, K), -3, 3, torch.float32, g)  # logical [i][k]
    Qm = _ints((J, K), -2, 2, torch.float32, g)  # logical [j][k]
    Pm[:, 0] += 1  # break symmetry
    ref = Pm.double() @ Qm.double().T
    P = (Pm.T if pks else Pm).contiguous().to(dt)
    Q = (Qm.T if qks else Qm).contiguous().to(dt)
    if layout == "wgrad":
        colsum = torch.zeros(I, dtype=torch.float32, device=U.DEV)
        out = U.gemm(mode, P, Q, I, J, K, pks, qks, out_f32=True, splitk=3, colsum=colsum)
        assert torch.equal(colsum.double(), Pm.double().sum(1)), "column sums (bias gradient)"
    else:
        bias = torch.arange(J, dtype=torch.float32, device=U.DEV) % 5 - 2
        out = U.gemm(mode, P, Q, I, J, K, pks, qks, bias=bias, out_f32=True)
        ref = ref + bias.double()[None]
    torch.cuda.synchronize()
    assert torch.equal(out.double(), ref), f"max |diff| {float((out.double() - ref).abs().max())}"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_gemm_random(mode, layout):
    pks, qks = LAYOUTS[layout]
    I, J, K = (272 if layout == "wgrad" else 270), 1920, 480  # a K-strided operand needs its extent in whole 16-byte chunks
    g = torch.Generator().manual_seed(5)
    dt = U.tdtype(mode)
    Pm = torch.randn((I, K), generator=g).to(U.DEV).to(dt)
    Qm = torch.randn((J, K), generator=g).to(U.DEV).to(dt)
    ref = Pm.double() @ Qm.double().T
    P = (Pm.T if pks else Pm).contiguous()
    Q = (Qm.T if qks else Qm).contiguous()
    out = U.gemm(mode, P, Q, I, J, K, pks, qks, out_f32=True, splitk=2)
    assert U.rel_err(out, ref) < 2e-5  # operands already rounded to the mode's type: only accumulation order differs
    if layout != "wgrad":
        out_t = U.gemm(mode, P, Q, I, J, K, pks, qks, out_f32=False)
        assert U.rel_err(out_t, ref) < (2e-5 if mode == "f32" else 8e-3)


def test_gemm_wgrad_accumulates():
    g = torch.Generator().manual_seed(1)
    Pm = _ints((64, 96), -2, 2, torch.float32, g)
    Qm = _ints((32, 96), -2, 2, torch.float32, g)
    out = torch.full((64, 32), 5.0, device=U.DEV)
    U.gemm("f32", Pm.T.contiguous(), Qm.T.contiguous(), 64, 32, 96, 1, 1, out_f32=True, out=out)
    assert torch.equal(out.double(), 5.0 + Pm.double() @ Qm.double().T)


def test_gemm_rejects_bad_arguments():
    a = torch.zeros((8, 30), device=U.DEV)
    with pytest.raises(RuntimeError, match="16-byte"):
        U.gemm("f32", a, a, 8, 8, 30, 0, 0)  # K = 30 floats is not a whole number of 16-byte chunks


# ----------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,T", [(2, 135), (1, 450), (3, 16), (1, 161)])
def test_attention_forward_backward(mode, B, T):
    H, dh = 6, 80
    g = torch.Generator().manual_seed(T)
    dt = U.tdtype(mode)
    qkv = (torch.randn((B * T, 3 * H * dh), generator=g) * 0.7).to(U.DEV).to(dt)
    do = torch.randn((B * T, H * dh), generator=g).to(U.DEV).to(dt)
    o, lse = U.attention_fwd(mode, qkv, B, T, H, dh)
    q64 = qkv.double().requires_grad_(True)
    ref = U.ref_attention(q64, B, T, H, dh)
    tol = 2e-5 if mode == "f32" else 1.5e-2
    assert U.rel_err(o, ref) < tol
    ref.backward(do.double())
    dqkv = U.attention_bwd(mode, qkv, o, do, lse, B, T, H, dh)
    assert U.rel_err(dqkv, q64.grad) < (5e-5 if mode == "f32" else 3e-2)
    # lse against a direct computation
    q, k, _ = qkv.double().reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    ref_lse = torch.logsumexp((q @ k.transpose(-1, -2)) * dh**-0.5, -1)
    assert float((lse.double() - ref_lse).abs().max()) < (1e-4 if mode == "f32" else 3e-2)


def test_attention_peaked_rows():
    """One key dominating a row (large logit spread) must not overflow / lose the row: online-softmax rescale path."""
    B, T, H, dh = 1, 450, 6, 80
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn((B * T, 3 * H * dh), generator=g)
    qkv[:, : H * dh] *= 6.0  # large queries -> spread of several hundred in the logits
    qkv = qkv.to(U.DEV)
    o, _ = U.attention_fwd("f32", qkv, B, T, H, dh)
    assert torch.isfinite(o).all()
    assert U.rel_err(o, U.ref_attention(qkv, B, T, H, dh)) < 5e-5


# ----------------------------------------------------------------------------------------------- layout / embeddings / LayerNorm
@pytest.mark.parametrize("cfg", [O.ds2(2), O.ds3(2)], ids=["ds2", "ds3"])
def test_patchify_bit_exact(cfg):
    from vit4hep_amd import _lib

    plan = _lib.Plan(cfg.shape, cfg.patch_shape, 46, 480, 2, 6, 1920)
    x = torch.randn((3, 1, *cfg.shape), generator=torch.Generator().manual_seed(0))
    xd = x.to(U.DEV)
    tok = torch.empty((3, cfg.T, cfg.P), device=U.DEV)
    _lib.check(_lib.load().v4h_op_patchify(plan.handle, _lib.ptr(xd), _lib.ptr(tok), 3, _lib.stream_ptr(), None))
    assert torch.equal(tok.cpu(), O.to_patches(x, cfg))
    back = torch.empty_like(xd)
    _lib.check(_lib.load().v4h_op_unpatchify(plan.handle, _lib.ptr(tok), _lib.ptr(back), 3, _lib.stream_ptr(), None))
    assert torch.equal(back.cpu(), x)
    assert torch.equal(back.cpu(), O.from_patches(O.to_patches(x, cfg), cfg))


@pytest.mark.parametrize("name,cfg", [("ds2_d2_b2", O.ds2(2)), ("ds3_d6_b1", O.ds3(6))])
def test_pos_embedding_vs_golden(name, cfg, golden):
    from vit4hep_amd import _lib

    g = golden(name)
    plan = _lib.Plan(cfg.shape, cfg.patch_shape, 46, 480, cfg.depth, 6, 1920)
    freqs = O.golden_fill(cfg)["pos_embed_freqs"].to(U.DEV)
    pe = torch.empty((cfg.T, 480), device=U.DEV)
    _lib.check(_lib.load().v4h_op_pos_embed(plan.handle, _lib.ptr(freqs), _lib.ptr(pe), _lib.stream_ptr(), None))
    assert float((pe.cpu().double() - torch.from_numpy(g["pos_embed"]).double()).abs().max()) < 2e-6


@pytest.mark.parametrize("mode", MODES)
def test_ln_modulate(mode):
    from vit4hep_amd import _lib

    B, T, D = 3, 135, 480
    g = torch.Generator().manual_seed(2)
    x = (torch.randn((B, T, D), generator=g) * 2 + 0.5).to(U.DEV)
    mod = torch.randn((B, 6 * D), generator=g).to(U.DEV) * 0.3
    u = torch.empty((B * T, D), dtype=U.tdtype(mode), device=U.DEV)
    mean = torch.empty(B * T, device=U.DEV)
    rstd = torch.empty(B * T, device=U.DEV)
    shift, scale = mod[:, 3 * D : 4 * D], mod[:, 4 * D : 5 * D]
    _lib.check(_lib.load().v4h_op_ln_modulate_fwd(_lib.MODES[mode], _lib.ptr(x), _lib.ptr(shift), _lib.ptr(scale), 6 * D, _lib.ptr(u), _lib.ptr(mean), _lib.ptr(rstd),
                                                  B, T, D, _lib.stream_ptr()))
    xd = x.double()
    ref = O.modulate(O.layernorm(xd), shift.double(), scale.double()).reshape(B * T, D)
    assert U.rel_err(u, ref) < (2e-6 if mode == "f32" else 6e-3)
    assert U.rel_err(mean, xd.mean(-1).flatten()) < 1e-5
