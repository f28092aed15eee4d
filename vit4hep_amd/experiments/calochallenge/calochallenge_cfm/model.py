"""Host mirror of the reference's CaloChallengeCFM (experiments/calochallenge/calochallenge_cfm/model.py:8-94).

Voxel grid <-> patch tokens around the ViT.  When ``net`` is this package's ViT the patching is fused into the
HIP forward/backward (voxels in, voxels out); ``to_patches`` / ``from_patches`` remain available as HIP kernels.
``CaloChallengeCFM_DS1`` (irregular layers: one patch segment per calorimeter layer) runs through the same kernels with an
index-map geometry (vit4hep_amd/patching.py).
"""

from __future__ import annotations

import torch

from ....autograd import _patchify, _unpatchify
from ....models.base_model import CFM
from ....models.segmented import SegmentedPatching
from ....models.segmented import unwrap as _unwrap
from ....nn.vit import ViT


class CaloChallengeCFM(CFM):
    def __init__(self, net, patch_shape, in_channels=1, time_distribution="uniform", trajectory="linear", odeint_kwargs=None, *args, **kwargs):
        super().__init__(None, time_distribution, trajectory, odeint_kwargs, *args, **kwargs)
        self.shape = [int(s) for s in self.shape]
        self.patch_shape = [int(p) for p in patch_shape]
        self.num_patches = [s // p for s, p in zip(self.shape, self.patch_shape)]
        self.in_channels = in_channels
        for i, (s, p) in enumerate(zip(self.shape, self.patch_shape)):
            assert s % p == 0, f"Input size ({s}) should be divisible by patch size ({p}) in axis {i}."
        if in_channels != 1:
            raise NotImplementedError("vit4hep_amd: in_channels != 1 is not on the shape-CFM path")
        self.net = net
        core = _unwrap(net)
        if not isinstance(core, ViT):
            raise TypeError("vit4hep_amd.CaloChallengeCFM needs a vit4hep_amd.nn.vit.ViT network: the path has no PyTorch fallback")
        core.set_geometry(self.shape, self.patch_shape)

    def _core(self):
        core = _unwrap(self.net)
        core.set_geometry(self.shape, self.patch_shape)
        return core

    def to_patches(self, x):
        """b c (l p1) (a p2) (r p3) -> b (l a r) (p1 p2 p3 c)    (reference model.py:54-60)"""
        return _patchify(self._core(), x.contiguous())

    def from_patches(self, x):
        """b (l a r) (p1 p2 p3 c) -> b c (l p1) (a p2) (r p3)    (reference model.py:40-52)"""
        return _unpatchify(self._core(), x.contiguous())

    def forward(self, x, t, c):
        """to_patches -> net -> from_patches (reference model.py:62-66), fused: self.net (possibly DDP-wrapped) gets voxels."""
        self._core()
        return self.net(x, t, c)

    @torch.inference_mode()
    def sample_batch(self, batch):
        """reference model.py:68-94"""
        x_T = torch.randn((batch.shape[0], self.in_channels, *self.shape), dtype=batch.dtype, device=batch.device)
        return self._sample_from(x_T, batch)


class CaloChallengeCFM_DS1(SegmentedPatching, CaloChallengeCFM):
    """Dataset-1 (photons / pions) shape model: per-layer segments with a shared patch shape (reference model.py:97-173)."""

    def __init__(self, net, list_shape, list_edges, patch_shape, in_channels=1, time_distribution="uniform", trajectory="linear", odeint_kwargs=None,
                 *args, **kwargs):
        CFM.__init__(self, None, time_distribution, trajectory, odeint_kwargs, *args, **kwargs)
        self.shape = [int(s) for s in self.shape]
        self.patch_shape = [int(p) for p in patch_shape]
        self.in_channels = in_channels
        self.num_patches = [s // p for s, p in zip(self.shape, self.patch_shape)]  # what the reference's base constructor leaves behind
        self._init_segments(net, list_shape, list_edges, [self.patch_shape] * len(list(list_shape)))
