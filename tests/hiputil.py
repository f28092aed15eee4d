"""Helpers for the -m gpu parity tests: thin ctypes calls into libvit4hep_hip.so on torch device tensors."""

import numpy as np
import torch

from vit4hep_amd import _lib

DEV = "cuda:0"


def tdtype(mode):
    return torch.bfloat16 if mode == "bf16" else torch.float32


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rms_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def gemm(mode, P, Q, I, J, K, pks, qks, bias=None, out_f32=False, splitk=1, colsum=None, out=None):
    """Out[i][j] = sum_k P.. Q..  through v4h_op_gemm.  P, Q are 2-D device tensors in their memory layout."""
    lib = _lib.load()
    if out is None:
        out = torch.zeros((I, J), dtype=torch.float32 if out_f32 else tdtype(mode), device=P.device)
    _lib.check(
        lib.v4h_op_gemm(_lib.MODES[mode], _lib.ptr(P), P.stride(0), int(pks), _lib.ptr(Q), Q.stride(0), int(qks), _lib.ptr(bias), _lib.ptr(out), out.stride(0),
                        int(out_f32), I, J, K, splitk, _lib.ptr(colsum), _lib.stream_ptr(P.device)),
        "v4h_op_gemm",
    )
    return out


def attention_fwd(mode, qkv, B, T, H, dh):
    lib = _lib.load()
    o = torch.empty((B * T, H * dh), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.v4h_op_attention_fwd(_lib.MODES[mode], _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, T, H, dh, _lib.stream_ptr(qkv.device)), "attention_fwd")
    return o, lse


def attention_bwd(mode, qkv, o, do, lse, B, T, H, dh):
    lib = _lib.load()
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    _lib.check(
        lib.v4h_op_attention_bwd(_lib.MODES[mode], _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(delta), _lib.ptr(dqkv), B, T, H, dh,
                                 _lib.stream_ptr(qkv.device)),
        "attention_bwd",
    )
    return dqkv


def ref_attention(qkv, B, T, H, dh):
    """fp64 torch reference on token-major qkv (B*T, 3*H*dh); returns o (B*T, H*dh)."""
    q, k, v = qkv.double().reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * dh**-0.5
    a = torch.softmax(s, -1)
    return (a @ v).transpose(1, 2).reshape(B * T, H * dh)


def build_net(cfg, mode):
    from vit4hep_amd import ViT

    return ViT({"dim": 3, "condition_dim": cfg.condition_dim, "hidden_dim": cfg.hidden_dim, "depth": cfg.depth, "num_heads": cfg.num_heads,
                "mlp_ratio": cfg.mlp_ratio, "patch_dim": cfg.P, "num_patches": [list(n) for n in cfg.seg_num_patches], "learn_pos_embed": True,
                "amd_mode": mode, "use_torch_sdpa": False, "pos_embedding_coords": "cylindrical"})


def build_models(cfg, mode, fill, device=DEV, kind="calochallenge"):
    """vit4hep_amd CFM wrapper (`kind`: calochallenge | ds1 | calogan | calohad | lemurs, the reference's wrapper classes) with the
    oracle's deterministic parameters loaded."""
    from vit4hep_amd import CaloChallengeCFM

    net = build_net(cfg, mode)
    common = dict(in_channels=1, odeint_kwargs={"method": "rk4", "options": {"step_size": 0.05}}, shape=list(cfg.shape))
    list_shape = [list(s) for s, _ in cfg.segments]
    list_edges = [int(np.prod(s)) for s, _ in cfg.segments]
    list_patch = [list(p) for _, p in cfg.segments]
    if kind == "calochallenge":
        model = CaloChallengeCFM(net, list(cfg.patch_shape), **common)
    elif kind == "ds1":
        from vit4hep_amd.experiments.calochallenge.calochallenge_cfm.model import CaloChallengeCFM_DS1

        model = CaloChallengeCFM_DS1(net, list_shape, list_edges, list_patch[0], **common)
    elif kind == "calogan":
        from vit4hep_amd.experiments.calogan.model import CaloGANCFM

        model = CaloGANCFM(net, list_shape, list_edges, list_patch, **common)
    elif kind == "calohad":
        from vit4hep_amd.experiments.calohadronic.model import CaloHadCFM

        model = CaloHadCFM(net, list_shape, list_edges, list_patch, **common)
    elif kind == "lemurs":
        from vit4hep_amd.experiments.lemurs.model import LEMURSCFM

        model = LEMURSCFM(net, list(cfg.patch_shape), **common)
    else:
        raise ValueError(kind)
    sd = model.state_dict()
    for k, v in fill.items():
        assert sd["net." + k].shape == v.shape, k
        sd["net." + k] = v.clone()
    model.load_state_dict(sd)
    model.device, model.dtype = torch.device(device), torch.float32
    return model.to(device, torch.float32)


def named_grads(model):
    return {k[4:]: p.grad for k, p in model.named_parameters()}
