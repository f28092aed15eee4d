"""CPU-only: the C-ABI library loads and exports every symbol include/vit4hep_hip.h declares; plan / error behaviour;
host mirror classes keep the reference's constructor, state-dict keys and initialisation; no compute without a GPU."""

import ctypes
import math
import os
import sys
import re

import numpy as np
import pytest
import torch

from oracle import vit_cfm_oracle as O
from vit4hep_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "vit4hep_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v4h_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vit4hep_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert _lib.load().v4h_abi_version() == _lib.ABI_VERSION == 11


def test_product_library_contains_no_ablation_kernels_and_refuses_to_select_one():
    """The ablation builds of the contraction kernels (some wrong by construction: no DMA after the first ring fill, one store per tile ...) and the
    tuning hook that selected them exist only under -DV4H_ABLATIONS.  A default build exports none of them, ignores an environment that asks for one and
    refuses any selection other than the exact kernels (0 automatic, 1 two-workgroup, 2 ring, 3 weight-stationary - round 5)."""
    import subprocess

    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "v4h_debug_" not in syms
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"dbg1", b"dbg2", b"dbg3", b"V4H_GEMM_CFG", b"V4H_GEMM_WCFG", b"V4H_THIRD_QUEUE", b"V4H_PP_FLAGS", b"V4H_GEMM_STRIPS"):
        assert needle not in blob, needle
    # mangled names of the ring kernel's instantiations: Gemm2Cfg<PKS, QKS, EPI, COLSUM, DBG, PP> - DBG must be 0 and PP true in every one
    for m in re.finditer(rb"8Gemm2CfgILb[01]ELb[01]ELi\d+ELb[01]ELi(\d+)ELb([01])EE", blob):
        assert m.group(1) == b"0" and m.group(2) == b"1", m.group(0)
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); print(lib.v4h_selected_contraction_kernel(), lib.v4h_select_contraction_kernel(4), "
            "lib.v4h_select_contraction_kernel(-1), lib.v4h_selected_contraction_kernel(), lib.v4h_select_contraction_kernel(2), lib.v4h_selected_contraction_kernel())")
    for env_val, first in (("3", 0), ("10", 0), ("0", 1), ("8", 2), ("-1", 0)):
        env = dict(os.environ, V4H_GEMM2=env_val)
        r = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], capture_output=True, text=True, env=env, check=True)
        assert r.stdout.split() == [str(first), "1", "1", str(first), "0", "2"], (env_val, r.stdout, r.stderr)
        assert ("ignored" in r.stderr) == (env_val in ("3", "10")), r.stderr


def test_compute_unit_reservation_accepts_only_whole_xcd_rows():
    """v4h_reserve_compute_units: the CUs the persistent grids leave to a communication kernel - a multiple of 8 (workgroup id % 8 must stay the XCD)
    in [0, 64]; anything else is refused and leaves the setting unchanged."""
    lib = _lib.load()
    assert lib.v4h_reserved_compute_units() == 0
    try:
        for bad in (-8, 4, 12, 72, 256):
            assert lib.v4h_reserve_compute_units(bad) == 1  # V4H_ERR_ARG
            assert lib.v4h_reserved_compute_units() == 0
        for ok in (16, 64, 8, 0):
            assert lib.v4h_reserve_compute_units(ok) == 0
            assert lib.v4h_reserved_compute_units() == ok
    finally:
        lib.v4h_reserve_compute_units(0)


def test_plan_inventory_matches_reference_state_dict():
    for cfg in (O.ds2(6), O.ds3(6), O.ds2(2)):
        plan = _lib.Plan(cfg.shape, cfg.patch_shape, cfg.condition_dim, cfg.hidden_dim, cfg.depth, cfg.num_heads, cfg.mlp_hidden)
        assert plan.shapes == [tuple(s) for s in O.param_shapes(cfg).values()]
        assert plan.num_stages == cfg.depth + 2
        assert plan.workspace_bytes(128, True) > plan.workspace_bytes(128, False) > 0
        assert plan.workspace_bytes(256, True) > plan.workspace_bytes(128, True)


def test_mapped_plan_inventory_for_the_multisegment_geometries():
    for cfg in (O.ds1_photons(6), O.ds1_pions(6), O.calogan(6), O.calohad(6)):
        plan = _lib.Plan(None, None, cfg.condition_dim, cfg.hidden_dim, cfg.depth, cfg.num_heads, cfg.mlp_hidden, mapped=(cfg.T, cfg.P, cfg.shape[0]))
        assert plan.mapped and plan.shapes == [tuple(s) for s in O.param_shapes(cfg).values()]
        assert plan.workspace_bytes(64, True) > plan.workspace_bytes(64, False) > 0
    with pytest.raises(RuntimeError, match="must be positive"):
        _lib.Plan(None, None, 6, 480, 2, 6, 1920, mapped=(0, 5, 440))
    with pytest.raises(RuntimeError, match="int32"):
        _lib.Plan(None, None, 6, 480, 2, 6, 1920, mapped=(88, 5, 1 << 31))


def test_segment_patch_map_is_the_reference_permutation():
    """The index table = what split / rearrange / cat does to voxel indices (oracle restatement of calochallenge_cfm/model.py:163-173)."""
    from vit4hep_amd.patching import multi_segment_meshgrid, segment_patch_map

    for cfg in (O.ds1_photons(1), O.ds1_pions(1), O.calogan(1), O.calohad(1)):
        shapes, patches = [s for s, _ in cfg.segments], [p for _, p in cfg.segments]
        pm, per_dim, per_layer, V = segment_patch_map(shapes, [math.prod(s) for s in shapes], patches)
        assert V == cfg.shape[0] and pm.shape == (cfg.T, cfg.P) and pm.dtype == np.int32
        assert per_dim == cfg.seg_num_patches and sum(per_layer) == cfg.T
        want = O.to_patches(torch.arange(V, dtype=torch.float32).reshape(1, 1, V), cfg)[0].numpy()
        assert np.array_equal(pm, want.astype(np.int32))
        assert np.array_equal(np.sort(pm.ravel()), np.arange(V))  # a permutation: every voxel exactly once
        for got, ref in zip(multi_segment_meshgrid(cfg.seg_num_patches), O.meshgrid_buffers(cfg)):
            assert np.array_equal(got, ref.numpy())
    with pytest.raises(AssertionError, match=r"Input size \(19\) should be divisible by patch size \(2\) in axis 2"):
        segment_patch_map([(1, 8, 5), (1, 16, 10), (1, 19, 10)], [40, 160, 190], [(1, 2, 5)] * 3)
    with pytest.raises(ValueError, match="list_edges"):
        segment_patch_map([(1, 8, 5)], [41], [(1, 1, 5)])
    with pytest.raises(ValueError, match="same patch_dim"):
        segment_patch_map([(1, 8, 5), (1, 8, 4)], [40, 32], [(1, 1, 5), (1, 1, 4)])


def test_multisegment_wrappers_mirror_the_reference_constructors():
    from tests import hiputil as U

    for cfg, kind, cls in ((O.ds1_photons(1), "ds1", "CaloChallengeCFM_DS1"), (O.calogan(1), "calogan", "CaloGANCFM"), (O.calohad(1), "calohad", "CaloHadCFM"),
                           (O.lemurs(1), "lemurs", "LEMURSCFM")):
        m = U.build_models(cfg, "f32", O.golden_fill(cfg), device="cpu", kind=kind)
        assert type(m).__name__ == cls and m.in_channels == 1 and m.shape == list(cfg.shape)
        core = m._core()
        assert core.num_tokens == cfg.T and core.voxel_shape() == tuple(cfg.shape)
        if cfg.segments:
            assert m.list_edges == [math.prod(s) for s, _ in cfg.segments] and m.num_patches_per_layer == [l * a * r for l, a, r in cfg.seg_num_patches]
            assert [tuple(n) for n in core.num_patches] == cfg.seg_num_patches and not core.map_has_holes()
            assert core._get_plan().mapped
        with pytest.raises(RuntimeError, match="MI355X"):  # still no CPU path
            m.forward(torch.zeros((1, 1, *cfg.shape)), torch.zeros(1, 1), torch.zeros(1, cfg.condition_dim))
    from vit4hep_amd.experiments.calochallenge.calochallenge_cfm.model import CaloChallengeCFM, CaloChallengeCFM_DS1

    assert issubclass(CaloChallengeCFM_DS1, CaloChallengeCFM)  # as in the reference (model.py:97)


def test_plan_rejects_what_the_reference_asserts():
    with pytest.raises(RuntimeError, match=r"Input size \(45\) should be divisible by patch size \(4\) in axis 0"):
        _lib.Plan((45, 16, 9), (4, 16, 1), 46, 480, 2, 6, 1920)  # calochallenge_cfm/model.py:33-36
    with pytest.raises(RuntimeError, match="divisible by num_heads"):
        _lib.Plan((45, 16, 9), (3, 16, 1), 46, 480, 2, 7, 1920)  # nn/vit.py:411
    with pytest.raises(RuntimeError, match="head_dim"):
        _lib.Plan((45, 16, 9), (3, 16, 1), 46, 384, 2, 6, 1536)


def _model(depth=2, **extra):
    from vit4hep_amd import CaloChallengeCFM, ViT

    param = {"dim": 3, "condition_dim": 46, "hidden_dim": 480, "out_channels": 1, "depth": depth, "num_heads": 6, "mlp_ratio": 4, "attn_drop": 0.0,
             "proj_drop": 0.0, "pos_embedding_coords": "cylindrical", "temperature": 10000, "learn_pos_embed": True, "causal_attn": False,
             "checkpoint_grads": False, "num_patches": [[15, 1, 9]], "patch_dim": 48, "use_torch_sdpa": False, "use_rotary_emb": False}
    param.update(extra)
    net = ViT(param)
    return CaloChallengeCFM(net, [3, 16, 1], in_channels=1, time_distribution="uniform", trajectory="linear",
                            odeint_kwargs={"method": "rk4", "options": {"step_size": 0.05}}, shape=[45, 16, 9])


def test_state_dict_keys_and_init_match_reference():
    torch.manual_seed(0)
    m = _model(2)
    cfg = O.ds2(2)
    want = ["net." + k for k in O.param_shapes(cfg)]
    got = [k for k, _ in m.named_parameters()]
    assert got == want
    assert [k for k, _ in m.named_buffers()] == ["net.pos_z", "net.pos_y", "net.pos_x"]
    assert sum(p.numel() for p in m.parameters()) == 9424928
    sd = m.state_dict()
    pz, py, px = O.meshgrid_buffers(cfg)
    assert torch.equal(sd["net.pos_z"], pz) and torch.equal(sd["net.pos_y"], py) and torch.equal(sd["net.pos_x"], px)
    for k, v in sd.items():  # nn/vit.py:164-183
        if k.endswith("bias") or "adaLN_modulation" in k or k.startswith("net.final_layer.linear"):
            assert float(v.abs().max()) == 0.0, k
        elif k.endswith("weight"):
            bound = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
            assert 0.5 * bound < float(v.abs().max()) <= bound * (1 + 1e-6), k
    # round trip with "module." prefixes stripped, as experiments/misc.py:65-71 does after DDP
    m2 = _model(2)
    ddp_sd = {k.replace("net.", "net.module.", 1): v for k, v in sd.items()}  # keys as saved from DDP(model.net)
    m2.load_state_dict({k.replace("module.", ""): v for k, v in ddp_sd.items()})
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_reference_attribute_surface():
    m = _model(2)
    net = m.net
    for name in ("x_embedder", "c_embedder", "t_embedder", "blocks", "final_layer", "pos_embed_freqs", "learn_pos_embed", "num_patches", "pos_z", "pos_y", "pos_x",
                 "create_meshgrid", "learnable_pos_embedding", "initialize_weights"):
        assert hasattr(net, name), name  # reached into by experiment_finetuning.py:80-196
    assert m.num_patches == [15, 1, 9] and m.patch_shape == [3, 16, 1] and m.in_channels == 1
    assert type(net).__name__ == "ViT"


def test_unsupported_options_fail_loudly():
    from vit4hep_amd import ViT

    base = {"hidden_dim": 480, "depth": 1, "num_heads": 6, "mlp_ratio": 4, "patch_dim": 48, "num_patches": [[15, 1, 9]]}
    for bad in ({"learn_pos_embed": False}, {"causal_attn": True}, {"attn_drop": 0.1}, {"num_patches": [[5, 9], [10, 9]]}, {"dim": 2}):
        with pytest.raises(NotImplementedError):
            ViT({**base, **bad})
    with pytest.raises(ValueError):
        ViT({**base, "amd_mode": "fp8"})
    from vit4hep_amd import CFM

    with pytest.raises(ValueError):
        CFM(None, trajectory="sine", shape=[1])  # models/base_model.py:186-190
    with pytest.raises(ValueError):
        CFM(None, time_distribution="beta", shape=[1])


def test_no_cpu_fallback():
    m = _model(1)
    m.device, m.dtype = torch.device("cpu"), torch.float32
    x, c, _ = O.synthetic_batch(O.ds2(1), 2, 0)
    with pytest.raises(RuntimeError, match="MI355X"):
        m.forward(x, torch.rand(2, 1), c)
    with pytest.raises(RuntimeError, match="MI355X"):
        m._batch_loss([x, c])
    from vit4hep_amd.trainer import CFMTrainer

    with pytest.raises(RuntimeError, match="MI355X"):
        CFMTrainer(m)


def test_solver_grid_and_lr_schedule_match_oracle():
    from vit4hep_amd.models.base_model import fixed_grid

    for step in (0.05, 0.25, 0.5, 0.3):
        assert np.array_equal(fixed_grid(0.0, 1.0, step), O.fixed_grid(0.0, 1.0, step).numpy())
    st = O.AdamWState(iterations=50)
    lr = lambda k: 1e-4 * 0.5 * (1.0 + math.cos(math.pi * k / 50))
    assert all(abs(st.lr_at(k) - lr(k)) < 1e-18 for k in range(60))


def test_dropin_aliases_reference_module_paths():
    import importlib
    import sys

    from vit4hep_amd import dropin

    saved = {k: sys.modules.get(k) for k in dropin.ALIASES}
    try:
        dropin.install()
        assert importlib.import_module("nn.vit").ViT is importlib.import_module("vit4hep_amd.nn.vit").ViT
        mod = importlib.import_module("experiments.calochallenge.calochallenge_cfm.model")
        assert mod.CaloChallengeCFM.__module__.startswith("vit4hep_amd")
        assert importlib.import_module("models.base_model").CFM.__module__.startswith("vit4hep_amd")
        assert mod.CaloChallengeCFM_DS1.__module__.startswith("vit4hep_amd")  # configs/model/cfm/cfm_ds1_photons.yaml:1
        for path, cls in (("experiments.calogan.model", "CaloGANCFM"), ("experiments.calohadronic.model", "CaloHadCFM"), ("experiments.lemurs.model", "LEMURSCFM")):
            assert getattr(importlib.import_module(path), cls).__module__.startswith("vit4hep_amd")  # configs/model/cfm_{calogan,calohad,lemurs}/*.yaml:1
        assert "nn.cfm.transformer_cfm" not in sys.modules or not sys.modules["nn.cfm.transformer_cfm"].__name__.startswith("vit4hep_amd")
        dropin.uninstall()
        dropin.install(energy_sampler=True)  # opt-in: the forward-only energy-model network
        assert importlib.import_module("nn.cfm.transformer_cfm").ParallelTransformer.__module__.startswith("vit4hep_amd")
    finally:
        dropin.uninstall()
        sys.modules.pop("nn.cfm", None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_driver_build_entry_point():
    """__graft_entry__.build(): compile (or reuse) the library, dlopen it, resolve every symbol, import every host class - on the CPU."""
    import importlib

    ge = importlib.import_module("__graft_entry__")
    ge.build()


def test_bench_gpus_n_launches_its_own_ranks_or_fails_loudly():
    """`python bench.py --gpus N` outside a launcher starts N child ranks itself (reference main.py:9-26 spawns its own); with fewer than N
    GPUs visible - as here - it must fail with a non-zero code and a message, not exit quietly without a JSON line."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    import torch

    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout, r.stderr[-2000:]
    else:
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and '{"metric"' not in r.stdout


def test_unbuilt_options_of_the_surface_raise_instead_of_being_ignored():
    """checkpoint_grads (nn/vit.py:201-202), optimizers / schedulers other than the fused AdamW + CosineAnnealingLR (base_experiment.py:329-431), an EMA decay
    outside [0, 1] (torch_ema's check): every option this build does not implement says so at construction."""
    from vit4hep_amd import ViT
    from vit4hep_amd.trainer import CFMTrainer

    base = {"hidden_dim": 480, "depth": 1, "num_heads": 6, "patch_dim": 48, "num_patches": [[15, 1, 9]]}
    with pytest.raises(NotImplementedError, match="checkpoint_grads"):
        ViT({**base, "checkpoint_grads": True})
    ViT({**base, "checkpoint_grads": False})
    for kw, pat in (({"optimizer": "Adam"}, "optimizer 'Adam'"), ({"optimizer": "Lion"}, "autograd route"), ({"scheduler": "OneCycleLR"}, "scheduler 'OneCycleLR'")):
        with pytest.raises(NotImplementedError, match=pat):
            CFMTrainer(None, **kw)
