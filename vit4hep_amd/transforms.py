"""Fused pre-/post-processing chain of the shape models (reference experiments/calochallenge/transforms.py).

The reference runs nine transform objects one after the other over the whole sample tensor (``for fn in transforms[::-1]: x, c =
fn(x, c, rev=True)``, experiments/calochallenge/experiment.py:190-223).  ``ShapeChain`` does the same arithmetic per shower in one HIP
kernel each way (csrc/v4h_transforms.hip).  ``ShapeChain.from_transforms(list_of_transform_objects)`` reads the constants off the
reference's own objects (duck-typed by class name), so an experiment swaps its loop for one call and keeps its YAML.
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch

from . import _lib

_ORDER = ["NormalizeByElayer", "ScaleTotalEnergy", "CutValues", "ExclusiveLogitTransform", "GlobalStandardizeFromFile", "LogEnergy", "ScaleEnergy",
          "AddFeaturesToCond", "Reshape"]


@dataclass
class ShapeChain:
    layer_boundaries: tuple
    shape: tuple
    eps: float = 1.0e-10
    norm_cut: float = 0.0
    factor: float = 0.35
    cut: float = 1.0e-7
    delta: float = 1.0e-6
    mean: float = 0.0
    std: float = 1.0
    alpha: float = 0.0
    e_min: float = 6.907755
    e_max: float = 13.815510

    def __post_init__(self):
        self.layer_boundaries = tuple(int(v) for v in self.layer_boundaries)
        self.shape = tuple(int(v) for v in self.shape)
        if list(self.layer_boundaries) != sorted(set(self.layer_boundaries)) or self.layer_boundaries[0] != 0:
            raise ValueError("layer_boundaries must start at 0 and increase strictly")
        if math.prod(self.shape) != self.n_voxels:
            raise ValueError(f"Reshape shape {self.shape} does not hold {self.n_voxels} voxels")
        self._bounds = {}

    @property
    def n_layers(self):
        return len(self.layer_boundaries) - 1

    @property
    def n_voxels(self):
        return self.layer_boundaries[-1]

    @classmethod
    def from_transforms(cls, transforms):
        """Build from the reference's transform objects, in the order of the YAML (calochallenge_ds2.yaml:15-28)."""
        names = [type(t).__name__ for t in transforms]
        if names != _ORDER:
            raise NotImplementedError(f"vit4hep_amd: fused chain implements {_ORDER}, got {names}")
        n, sc, cu, lg, st, le, se, af, rs = transforms
        if not getattr(lg, "rescale", False) or getattr(lg, "exclusions", None) is not None:
            raise NotImplementedError("vit4hep_amd: ExclusiveLogitTransform must have rescale=True and no exclusions")
        if not getattr(st, "written", False):
            raise NotImplementedError("vit4hep_amd: GlobalStandardizeFromFile must carry stored statistics (means.npy / stds.npy)")
        bounds = tuple(int(v) for v in n.layer_boundaries)
        if int(sc.n_layers) != len(bounds) - 1 or int(cu.n_layers) != len(bounds) - 1 or int(af.split_index) != bounds[-1]:
            raise ValueError("n_layers / split_index of the transforms do not agree with the layer boundaries")
        return cls(bounds, tuple(rs.shape), float(n.eps), float(n.cut), float(sc.factor), float(cu.cut), float(lg.delta), float(st.mean), float(st.std),
                   float(le.alpha), float(se.e_min), float(se.e_max))

    # ------------------------------------------------------------------ HIP plumbing
    def _spec(self):
        return _lib.V4HChainSpec(self.n_layers, self.n_voxels, self.eps, self.norm_cut, self.factor, self.cut, self.delta, self.mean, self.std, self.alpha,
                                 self.e_min, self.e_max)

    def _dev_bounds(self, device):
        key = str(device)
        if key not in self._bounds:
            self._bounds[key] = torch.tensor(self.layer_boundaries, dtype=torch.int32, device=device)
        return self._bounds[key]

    def preprocess(self, showers, energy):
        """showers (B, n_voxels) energies, incident energy (B, 1) -> x (B, *shape), c (B, n_layers + 1) = [u_0 .. u_{n-1} | energy]."""
        showers = _lib.require_cuda(showers, "showers")
        energy = _lib.require_cuda(energy, "energy").reshape(-1)
        B = showers.shape[0]
        if tuple(showers.shape) != (B, self.n_voxels) or energy.numel() != B:
            raise RuntimeError(f"bad shapes: showers {tuple(showers.shape)}, energy {tuple(energy.shape)} for {self.n_voxels} voxels")
        x = torch.empty((B, *self.shape), dtype=torch.float32, device=showers.device)
        c = torch.empty((B, self.n_layers + 1), dtype=torch.float32, device=showers.device)
        spec = self._spec()
        _lib.check(_lib.load().v4h_shape_preprocess(C.byref(spec), _lib.ptr(self._dev_bounds(showers.device)), _lib.ptr(showers), _lib.ptr(energy), _lib.ptr(x),
                                                    _lib.ptr(c), B, _lib.stream_ptr(showers.device)), "v4h_shape_preprocess")
        return x, c

    def postprocess(self, samples, cond):
        """samples (B, *shape), cond (B, n_layers + 1) -> showers (B, n_voxels), incident energy (B, 1)."""
        samples = _lib.require_cuda(samples, "samples")
        cond = _lib.require_cuda(cond, "conditions")
        B = samples.shape[0]
        if samples.numel() != B * self.n_voxels or tuple(cond.shape) != (B, self.n_layers + 1):
            raise RuntimeError(f"bad shapes: samples {tuple(samples.shape)}, conditions {tuple(cond.shape)} for {self.n_voxels} voxels, {self.n_layers} layers")
        showers = torch.empty((B, self.n_voxels), dtype=torch.float32, device=samples.device)
        energy = torch.empty((B, 1), dtype=torch.float32, device=samples.device)
        spec = self._spec()
        _lib.check(_lib.load().v4h_shape_postprocess(C.byref(spec), _lib.ptr(self._dev_bounds(samples.device)), _lib.ptr(samples), _lib.ptr(cond), _lib.ptr(showers),
                                                     _lib.ptr(energy), B, _lib.stream_ptr(samples.device)), "v4h_shape_postprocess")
        return showers, energy
