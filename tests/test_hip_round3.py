"""GPU parity tests (through the C ABI) of what round 3 added:

  * the whole-K kernels for the batch-row contractions of the conditioning path (v4h_gemm_small.h: embedder MLPs nn/vit.py:77-81,361-365, their input
    gradients and their weight gradients over B tokens) - exact-integer operator test + every gradient tensor of an update step at a batch that
    reaches them (B % 32 == 0) against the oracle's autograd;
  * the fused GELU / DGELU operator entry points (timm Mlp, nn/vit.py:312-322) on both product kernels;
  * the instruction-lean single-chunk attention forward (v4h_attention_dense.h, nn/vit.py:425-451) where a persistent workgroup walks several
    (batch, head) items - the case a two-sample test never reaches - for every tile count it is built for (6 .. 10 tiles: T = 81 .. 160; ds1 photons 88,
    ds1 pions 125, ds2 / LEMURS 135).
"""

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from tests import hiputil as U
from vit4hep_amd import _lib

pytestmark = pytest.mark.gpu


def _ints(shape, gen, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=gen, device=U.DEV).to(torch.bfloat16)


def test_small_k_weight_gradient_exact():
    """dW += dY^T X over K <= 512 tokens with no K split (v4h_smallk_wgrad_kernel): integer operands, so the f32 sums are exact; the result is
    accumulated onto what the gradient tensor held; the bias gradient (column sums of dY) likewise."""
    gen = torch.Generator(device=U.DEV).manual_seed(31)
    for K, I, J in ((128, 480, 480), (32, 480, 64), (256, 72, 32), (512, 480, 256), (96, 1000, 160)):
        P, Q = _ints((K, I), gen), _ints((K, J), gen)
        out = torch.full((I, J), 2.0, device=U.DEV)
        cs = torch.full((I,), -1.0, device=U.DEV)
        U.gemm("bf16", P, Q, I, J, K, 1, 1, out_f32=True, splitk=1, colsum=cs, out=out)
        assert torch.equal(out, P.float().t() @ Q.float() + 2.0), (K, I, J)
        assert torch.equal(cs, P.float().sum(0) - 1.0), (K, I, J)


def test_update_step_gradients_at_batch_32_match_the_oracle():
    """B = 32 is the smallest batch at which the embedder weight gradients take the whole-K kernel (K = B must be a multiple of 32); the embedder MLPs
    and their input gradients take theirs at every batch.  bf16 mode (the kernels are bf16-only): every gradient tensor against the oracle's autograd."""
    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    x, c, g = O.synthetic_batch(cfg, 32, 21)
    t, x0 = O.synthetic_noise(cfg, 32, g)
    model = U.build_models(cfg, "bf16", fill)
    loss = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
    loss.backward()
    ref_loss, _, ref = O.loss_and_grads(fill, x, c, t, x0, cfg)
    assert abs(loss.item() - ref_loss.item()) / ref_loss.item() < 3e-2
    grads = U.named_grads(model)
    for k, r in ref.items():
        if k.endswith("attn.qkv.bias"):
            continue  # (the key third is analytically zero: rounding noise on both sides; docs/history_r01-r04.md section 2)
        assert U.rms_err(grads[k], r) < 3e-2, (k, U.rms_err(grads[k], r))


def _gelu_ref(x):
    u = 0.7978845608028654 * (x + 0.044715 * x**3)
    t = torch.tanh(u)
    return 0.5 * x * (1 + t), 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * 0.7978845608028654 * (1 + 3 * 0.044715 * x * x)


@pytest.mark.parametrize("kernel", [_lib.KERNEL_TWO_WG, _lib.KERNEL_RING])
def test_gelu_and_dgelu_operators(kernel):
    """fc1 + tanh-GELU (value and saved derivative) and the fc2 input gradient times that derivative, each as ONE contraction (v4h_op_gemm_gelu /
    v4h_op_gemm_dgelu), on the two-workgroup kernel and on the ring kernel: against torch in f64 on the same bf16 operands; a ragged last row tile."""
    lib = _lib.load()
    gen = torch.Generator(device=U.DEV).manual_seed(12)
    I, D, M = 2300, 480, 1920
    s = _lib.stream_ptr(U.DEV)
    _lib.check(lib.v4h_select_contraction_kernel(kernel), "select")
    try:
        x = torch.randn((I, D), generator=gen, device=U.DEV).to(torch.bfloat16)
        W1 = (torch.randn((M, D), generator=gen, device=U.DEV) * D**-0.5).to(torch.bfloat16)
        b1 = torch.randn(M, generator=gen, device=U.DEV) * 0.1
        h = torch.zeros((I, M), device=U.DEV, dtype=torch.bfloat16)
        dh = torch.zeros_like(h)
        _lib.check(lib.v4h_op_gemm_gelu(_lib.MODES["bf16"], _lib.ptr(x), D, _lib.ptr(W1), D, _lib.ptr(b1), _lib.ptr(h), M, _lib.ptr(dh), M, I, M, D, s), "gemm_gelu")
        y, dy = _gelu_ref(x.double() @ W1.double().t() + b1.double())
        assert U.rel_err(h, y) < 6e-3 and U.rms_err(h, y) < 3e-3
        assert U.rel_err(dh, dy) < 6e-3 and U.rms_err(dh, dy) < 3e-3
        h2 = torch.zeros_like(h)  # inference form: no derivative output
        _lib.check(lib.v4h_op_gemm_gelu(_lib.MODES["bf16"], _lib.ptr(x), D, _lib.ptr(W1), D, _lib.ptr(b1), _lib.ptr(h2), M, None, M, I, M, D, s), "gemm_gelu")
        assert U.rel_err(h2, y) < 6e-3
        g = torch.randn((I, D), generator=gen, device=U.DEV).to(torch.bfloat16)
        W2 = (torch.randn((D, M), generator=gen, device=U.DEV) * D**-0.5).to(torch.bfloat16)
        out = torch.zeros((I, M), device=U.DEV, dtype=torch.bfloat16)
        _lib.check(lib.v4h_op_gemm_dgelu(_lib.MODES["bf16"], _lib.ptr(g), D, _lib.ptr(W2), M, _lib.ptr(dh), M, _lib.ptr(out), M, I, M, D, s), "gemm_dgelu")
        ref = (g.double() @ W2.double()) * dh.double()
        assert U.rel_err(out, ref) < 6e-3 and U.rms_err(out, ref) < 3e-3
    finally:
        lib.v4h_select_contraction_kernel(_lib.KERNEL_AUTO)


@pytest.mark.parametrize("B,T", [(128, 135), (40, 160), (48, 150), (9, 129), (64, 88), (70, 125), (50, 100), (90, 81)])
def test_single_chunk_attention_with_several_items_per_workgroup(B, T):
    """256 persistent workgroups, B * 6 (batch, head) items: up to three items per workgroup through the double-buffered images (the second and third
    items are where a stale image, a late DMA or a mis-scheduled MFMA pair would show), rows >= T of the last tile masked / dropped, both tile counts."""
    H, dh = 6, 80
    gen = torch.Generator(device=U.DEV).manual_seed(B * 1000 + T)
    qkv = torch.randn((B * T, 3 * H * dh), generator=gen, device=U.DEV).to(torch.bfloat16)
    o, lse = U.attention_fwd("bf16", qkv, B, T, H, dh)
    q, k, v = [z.reshape(B, T, H, dh).transpose(1, 2).double() for z in qkv.reshape(B * T, 3, H * dh).unbind(1)]
    sc = q @ k.transpose(-1, -2) / dh**0.5
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * T, H * dh)
    err = (o.double() - ref).abs().reshape(B, T, H, dh).amax(dim=(1, 3))  # per (batch, head) item
    assert float(err.max()) < 2e-2, (int((err > 2e-2).sum()), "items wrong")
    assert float((lse.double() - torch.logsumexp(sc, -1)).abs().max()) < 5e-3
    # structured case: scores linear in the key index, v = one-hot(key mod 80) - every key's probability is visible in the output
    x = torch.zeros((B, T, 3, H, dh), device=U.DEV)
    x[:, :, 0, :, 0] = 4.0
    x[:, :, 1, :, 0] = (torch.arange(T, device=U.DEV).float() / 16)[None, :, None]
    x[:, :, 1, :, 1:] = torch.randn((B, T, H, dh - 1), generator=gen, device=U.DEV)
    oh = torch.zeros((T, dh), device=U.DEV)
    oh[torch.arange(T), torch.arange(T) % dh] = 1
    x[:, :, 2] = oh[None, :, None, :]
    o, _ = U.attention_fwd("bf16", x.reshape(B * T, 3 * H * dh).to(torch.bfloat16), B, T, H, dh)
    p = torch.softmax(4.0 * torch.arange(T, device=U.DEV).float() / 16 / dh**0.5, 0)
    want = torch.zeros(dh, device=U.DEV).index_add_(0, torch.arange(T, device=U.DEV) % dh, p)
    assert float((o.float().reshape(B, T, H, dh) - want).abs().max()) < 4e-3


@pytest.mark.parametrize("B,T", [(128, 135), (40, 160), (48, 150), (9, 129), (44, 450), (43, 470), (45, 369)])
def test_attention_backward_with_several_items_per_workgroup(B, T):
    """The persistent backward kernels - single-chunk (T = 129 .. 160) and whole-item images (T = 369 .. 480: attn_bwd_dq_img / attn_bwd_dkv_img of round 3, half
    an item per unit) - where a workgroup walks several items / units: dq, dk, dv of every (batch, head) item against
    torch autograd in f64 on the same bf16 inputs, per item (a stale image, a late DMA or a result stored for the wrong item shows as whole items wrong),
    and rows beyond the sequence untouched."""
    H, dh = 6, 80
    gen = torch.Generator(device=U.DEV).manual_seed(B * 1000 + T)
    qkv = (torch.randn((B * T, 3 * H * dh), generator=gen, device=U.DEV) * 0.7).to(torch.bfloat16)
    do = torch.randn((B * T, H * dh), generator=gen, device=U.DEV).to(torch.bfloat16)
    o, lse = U.attention_fwd("bf16", qkv, B, T, H, dh)
    q64 = qkv.double().requires_grad_(True)
    ref = U.ref_attention(q64, B, T, H, dh)
    assert U.rel_err(o, ref) < 1.5e-2
    ref.backward(do.double())
    dqkv = U.attention_bwd("bf16", qkv, o, do, lse, B, T, H, dh)
    assert bool(torch.isfinite(dqkv.float()).all())
    err = (dqkv.double() - q64.grad).reshape(B, T, 3, H, dh)
    scale = q64.grad.abs().reshape(B, T, 3, H, dh).amax(dim=(1, 4), keepdim=True)  # per (sample, q/k/v, head)
    worst = (err.abs() / scale).amax(dim=(1, 4))
    assert float(worst.max()) < 4e-2, (int((worst > 4e-2).sum()), "of", worst.numel(), "(item, tensor) slices wrong")
    assert U.rel_err(dqkv, q64.grad) < 3e-2


def test_reserving_compute_units_changes_no_result():
    """v4h_reserve_compute_units(n) shrinks the persistent grids (contractions on both kernels, single-chunk attention forward and backward) to 256 - n
    workgroups; the tile walk redistributes and no value may change.  Loss and every gradient of an update step at n = 0, 0, 16, 64."""
    lib = _lib.load()
    cfg = O.ds2(2)
    fill = O.golden_fill(cfg)
    x, c, g = O.synthetic_batch(cfg, 24, 5)
    t, x0 = O.synthetic_noise(cfg, 24, g)
    outs = []
    try:
        for n in (0, 0, 16, 64):
            _lib.check(lib.v4h_reserve_compute_units(n), "reserve")
            model = U.build_models(cfg, "bf16", fill)
            loss = model._loss_from_noise(x.to(U.DEV), c.to(U.DEV), t.to(U.DEV), x0.to(U.DEV))
            loss.backward()
            outs.append({"loss": loss.detach().clone(), **{k: v.clone() for k, v in U.named_grads(model).items()}})
    finally:
        lib.v4h_reserve_compute_units(0)
    # The token-stream contractions and attention are deterministic; the bias gradients (column sums), what hangs off the conditioning path and the loss
    # are summed with f32 atomics, whose order differs from run to run whatever n is: those are held to rounding noise instead.
    stream = [k for k in outs[0] if k.startswith("blocks.") and (".attn." in k or ".mlp." in k) and k.endswith(".weight")]
    assert len(stream) == 8
    for k in stream:
        assert torch.equal(outs[0][k], outs[1][k]), k  # (the premise: deterministic at fixed n)
    for o in outs[2:]:
        for k, v in o.items():
            if k in stream:
                assert torch.equal(v, outs[0][k]), k
            else:
                assert U.rms_err(v, outs[0][k]) < 1e-3, (k, U.rms_err(v, outs[0][k]))
