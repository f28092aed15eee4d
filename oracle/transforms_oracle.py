"""CPU oracle of the shape models' pre-/post-processing chain: the transforms the reference applies to showers
before training and, reversed, to the samples the shape model returns (experiments/calochallenge/experiment.py:52-75,190-223;
configs/calochallenge/cfm/calochallenge_ds2.yaml:15-28).

TEST INFRASTRUCTURE ONLY - imported by tests/, oracle/make_golden.py and the benchmark tools' CPU-baseline legs, never by the
product path (vit4hep_amd/).

Restates experiments/calochallenge/transforms.py as pure functions on torch CPU tensors (no in-place surprises):
  NormalizeByElayer 331-397, ScaleTotalEnergy 184-202, CutValues 291-311, ExclusiveLogitTransform 227-254 (+ logit 11-18),
  GlobalStandardizeFromFile 21-64 (statistics given), LogEnergy 149-164, ScaleEnergy 205-224, AddFeaturesToCond 130-146,
  Reshape 314-328.
Pinned by tests/golden/transforms_*.npz, produced by running the reference's own classes (oracle/make_golden.py).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class ChainSpec:
    """The ds2/ds3 (and ds1) shape-model chain of configs/calochallenge/cfm/calochallenge_ds2.yaml:15-28."""

    layer_boundaries: tuple          # voxel index where each layer starts, plus the total (NormalizeByElayer.layer_boundaries)
    shape: tuple = (1, 45, 16, 9)    # Reshape
    eps: float = 1.0e-10             # NormalizeByElayer eps
    norm_cut: float = 0.0            # NormalizeByElayer cut
    factor: float = 0.35             # ScaleTotalEnergy
    cut: float = 1.0e-7              # CutValues
    delta: float = 1.0e-6            # ExclusiveLogitTransform, rescale=True
    mean: float = 0.0                # GlobalStandardizeFromFile
    std: float = 1.0
    alpha: float = 0.0               # LogEnergy
    e_min: float = 6.907755          # ScaleEnergy
    e_max: float = 13.815510

    @property
    def n_layers(self):
        return len(self.layer_boundaries) - 1

    @property
    def n_voxels(self):
        return self.layer_boundaries[-1]


def ds2_spec(**kw):
    return ChainSpec(layer_boundaries=tuple(range(0, 6481, 144)), shape=(1, 45, 16, 9), **kw)


def preprocess(showers, energy, s: ChainSpec):
    """Forward chain: showers (B, n_voxels) in energy units, incident energy (B, 1) -> x (B, *shape), c (B, n_layers + 1)."""
    x = showers.clone()
    B = x.shape[0]
    # NormalizeByElayer forward (transforms.py:377-396)
    layer_Es = []
    for a, b in zip(s.layer_boundaries[:-1], s.layer_boundaries[1:]):
        e = x[:, a:b].sum(dim=1, keepdim=True)
        x[:, a:b] = x[:, a:b] / (e + s.eps)
        layer_Es.append(e)
    layer_Es = torch.cat(layer_Es, dim=1)
    extra = [layer_Es.sum(dim=1, keepdim=True) / energy]
    for L in range(s.n_layers - 1):
        remaining = layer_Es[:, L:].sum(dim=1, keepdim=True)
        extra.append(layer_Es[:, [L]] / (remaining + s.eps))
    x = torch.cat([x] + extra, dim=1)
    # ScaleTotalEnergy forward (transforms.py:197-202): only u_0
    x[:, -s.n_layers] = x[:, -s.n_layers] * s.factor
    # CutValues forward: identity (transforms.py:309-310)
    # ExclusiveLogitTransform forward, rescale=True (transforms.py:11-18,246-249)
    x = torch.logit(x * (1 - 2 * s.delta) + s.delta)
    # GlobalStandardizeFromFile forward with stored statistics (transforms.py:62)
    x = (x - s.mean) / s.std
    # LogEnergy, ScaleEnergy forward (transforms.py:163,221-222)
    c = (torch.log(energy + s.alpha) - s.e_min) / (s.e_max - s.e_min)
    # AddFeaturesToCond forward (transforms.py:143-145): the u's move in front of the condition
    c = torch.cat([x[:, s.n_voxels:], c], dim=1)
    x = x[:, : s.n_voxels]
    return x.reshape(B, *s.shape), c


def postprocess(samples, cond, s: ChainSpec):
    """Reverse chain: samples (B, *shape), cond (B, n_layers + 1) -> showers (B, n_voxels), incident energy (B, 1)."""
    B = samples.shape[0]
    x = samples.reshape(B, -1)                                  # Reshape rev (transforms.py:324-325)
    energy, us = cond[:, -1:], cond[:, :-1]                     # AddFeaturesToCond rev (transforms.py:140-142)
    x = torch.cat([x, us], dim=1)
    energy = energy * (s.e_max - s.e_min) + s.e_min             # ScaleEnergy rev (transforms.py:218-220)
    energy = torch.exp(energy) - s.alpha                        # LogEnergy rev (transforms.py:160-161)
    x = x * s.std + s.mean                                      # GlobalStandardizeFromFile rev (transforms.py:52)
    x = (torch.sigmoid(x) - s.delta) / (1 - 2 * s.delta)        # ExclusiveLogitTransform rev, rescale (transforms.py:13-15,242-243)
    if s.cut:                                                   # CutValues rev (transforms.py:303-308): voxels only
        vox = x[:, : -s.n_layers]
        x = torch.cat([torch.where(vox <= s.cut, torch.zeros_like(vox), vox), x[:, -s.n_layers :]], dim=1)
    x = x.clone()
    x[:, -s.n_layers] = x[:, -s.n_layers] / s.factor            # ScaleTotalEnergy rev (transforms.py:198-199)
    # NormalizeByElayer rev (transforms.py:345-375)
    us = x[:, -s.n_layers :].clone()
    us[:, 1:] = torch.clip(us[:, 1:], 0.0, 1.0)
    vox = x[:, : -s.n_layers]
    total = energy.flatten() * us[:, 0]
    cum = torch.zeros_like(total)
    layer_E = []
    for i in range(s.n_layers - 1):
        e = (total - cum) * us[:, i + 1]
        layer_E.append(e)
        cum = cum + e
    layer_E.append(total - cum)
    out = torch.zeros_like(vox)
    for L, (a, b) in enumerate(zip(s.layer_boundaries[:-1], s.layer_boundaries[1:])):
        layer = vox[:, a:b] / (vox[:, a:b].sum(-1, keepdim=True) + s.eps)
        layer = torch.where(layer <= s.norm_cut, torch.zeros_like(layer), layer)
        out[:, a:b] = layer * layer_E[L][:, None]
    return out, energy
