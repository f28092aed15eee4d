"""Pins the pre-/post-processing oracle (oracle/transforms_oracle.py) against vectors produced by the reference's own transform
classes (oracle/make_golden.py), and checks the host-side ShapeChain logic.  CPU only."""

import numpy as np
import pytest
import torch

from oracle import transforms_oracle as TO

SPECS = {"transforms_ds2_b4": TO.ds2_spec(mean=-1.7, std=2.9),
         "transforms_ds1ph_b6": TO.ChainSpec(layer_boundaries=(0, 8, 168, 358, 363, 368), shape=(368,), mean=-0.8, std=3.3, factor=0.5, cut=1e-6)}


@pytest.mark.parametrize("name", list(SPECS))
def test_oracle_is_bit_identical_to_the_reference_classes(name, golden):
    g, s = golden(name), SPECS[name]
    assert tuple(int(v) for v in g["bounds"]) == s.layer_boundaries
    x, c = TO.preprocess(torch.from_numpy(g["showers"]), torch.from_numpy(g["energy"]), s)
    assert np.array_equal(x.numpy(), g["x"]) and np.array_equal(c.numpy(), g["c"])
    sh, e = TO.postprocess(torch.from_numpy(g["samples"]), torch.from_numpy(g["cond"]), s)
    assert np.array_equal(sh.numpy(), g["post_showers"]) and np.array_equal(e.numpy(), g["post_energy"])
    sh, e = TO.postprocess(torch.from_numpy(g["x"]), torch.from_numpy(g["c"]), s)
    assert np.array_equal(sh.numpy(), g["roundtrip_showers"]) and np.array_equal(e.numpy(), g["roundtrip_energy"])
    # the chain is (nearly) invertible where it is meant to be
    keep = g["showers"] > 1e-5 * g["showers"].max()
    assert np.abs(g["roundtrip_showers"][keep] / g["showers"][keep] - 1).max() < 1e-4
    assert np.abs(g["roundtrip_energy"] / g["energy"] - 1).max() < 1e-5


def _fake(name, **attrs):
    """An object whose class carries the reference's class name (the reference itself is not importable on the GPU box)."""
    return type(name, (), attrs)()


def test_shape_chain_reads_the_reference_transform_objects():
    from vit4hep_amd.transforms import ShapeChain

    bounds = np.arange(0, 6481, 144)
    objs = [_fake("NormalizeByElayer", layer_boundaries=bounds, n_layers=45, eps=1e-10, cut=0.0), _fake("ScaleTotalEnergy", factor=0.35, n_layers=45),
            _fake("CutValues", cut=1e-7, n_layers=45), _fake("ExclusiveLogitTransform", delta=1e-6, rescale=True, exclusions=None),
            _fake("GlobalStandardizeFromFile", mean=torch.tensor(-1.7), std=torch.tensor(2.9), written=True), _fake("LogEnergy", alpha=0.0),
            _fake("ScaleEnergy", e_min=6.907755, e_max=13.815510), _fake("AddFeaturesToCond", split_index=6480), _fake("Reshape", shape=torch.Size([1, 45, 16, 9]))]
    ch = ShapeChain.from_transforms(objs)
    s = SPECS["transforms_ds2_b4"]
    assert ch.layer_boundaries == s.layer_boundaries and ch.shape == s.shape and ch.n_layers == 45 and ch.n_voxels == 6480
    for k in ("eps", "norm_cut", "factor", "cut", "delta", "alpha", "e_min", "e_max"):
        assert getattr(ch, k) == pytest.approx(getattr(s, k))
    assert ch.mean == pytest.approx(-1.7) and ch.std == pytest.approx(2.9)
    with pytest.raises(NotImplementedError, match="fused chain implements"):
        ShapeChain.from_transforms(objs[:3] + objs[4:])
    objs[3] = _fake("ExclusiveLogitTransform", delta=1e-6, rescale=False, exclusions=None)
    with pytest.raises(NotImplementedError, match="rescale"):
        ShapeChain.from_transforms(objs)
    with pytest.raises(ValueError, match="layer_boundaries"):
        ShapeChain((0, 10, 10, 20), (20,))
    with pytest.raises(ValueError, match="Reshape"):
        ShapeChain((0, 10, 20), (21,))
    with pytest.raises(RuntimeError, match="MI355X"):  # no CPU path
        ch.postprocess(torch.zeros((2, 1, 45, 16, 9)), torch.zeros((2, 46)))
