"""Shared host logic of the multi-segment CFM wrappers (CaloChallengeCFM_DS1, CaloGANCFM, CaloHadCFM).

The reference repeats the same constructor bookkeeping and split / rearrange / cat patching in three classes
(calochallenge_cfm/model.py:97-173, experiments/calogan/model.py:8-86, experiments/calohadronic/model.py:8-86); here the
patching is one index table (vit4hep_amd/patching.py) consumed by the HIP kernels, and the wrappers only differ in whether the
patch shape is shared or per segment.
"""

from __future__ import annotations

import torch

from ..autograd import _patchify, _unpatchify
from ..nn.vit import ViT
from ..patching import segment_patch_map


def unwrap(net):
    return net.module if hasattr(net, "module") and not isinstance(net, ViT) else net


class SegmentedPatching:
    """Mixin: expects ``self.in_channels``; call ``_init_segments`` after ``CFM.__init__``."""

    def _init_segments(self, net, list_shape, list_edges, list_patch_shape):
        if self.in_channels != 1:
            raise NotImplementedError("vit4hep_amd: in_channels != 1 is not on the shape-CFM path")
        self.list_shape = [list(int(v) for v in s) for s in list_shape]
        self.list_edges = [int(e) for e in list_edges]
        self._list_patch_shape = [list(int(v) for v in p) for p in list_patch_shape]
        assert len(self.list_shape) == len(self._list_patch_shape), "list_shape and list_patch_shape must have the same length"
        pmap, per_dim, per_layer, voxels = segment_patch_map(self.list_shape, self.list_edges, self._list_patch_shape)
        self.num_patches_per_dim = per_dim
        self.num_patches_per_layer = per_layer
        self._patch_map, self._voxels = pmap, voxels
        self.net = net
        core = unwrap(net)
        if not isinstance(core, ViT):
            raise TypeError(f"vit4hep_amd.{type(self).__name__} needs a vit4hep_amd.nn.vit.ViT network: the path has no PyTorch fallback")
        core.num_patches = self.num_patches_per_dim  # as the reference does (e.g. calohadronic/model.py:57)
        core.set_patch_map(pmap, voxels)

    def _core(self):
        core = unwrap(self.net)
        core.set_patch_map(self._patch_map, self._voxels)
        return core

    def to_patches(self, x):
        """(B, C, n_voxels) -> (B, T, P): split by list_edges, per-segment rearrange, cat - as one HIP gather."""
        return _patchify(self._core(), x.contiguous())

    def from_patches(self, x):
        """(B, T, P) -> (B, C, n_voxels), the inverse scatter."""
        return _unpatchify(self._core(), x.contiguous())

    def forward(self, x, t, c):
        """to_patches -> net -> from_patches, fused: self.net (possibly DDP-wrapped) gets the flat voxels."""
        self._core()
        return self.net(x, t, c)

    @torch.inference_mode()
    def sample_batch(self, batch):
        x_T = torch.randn((batch.shape[0], self.in_channels, *self.shape), dtype=batch.dtype, device=batch.device)
        return self._sample_from(x_T, batch)
