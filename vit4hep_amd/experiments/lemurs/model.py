"""Host mirror of the reference's experiments/lemurs/model.py: LEMURSCFM (regular 45 x 16 x 9 grid, 53 conditions)."""

from __future__ import annotations

import torch

from ...autograd import _patchify, _unpatchify
from ...models.base_model import CFM
from ...models.segmented import unwrap
from ...nn.vit import ViT


class LEMURSCFM(CFM):
    """Same patching as CaloChallengeCFM; ``_batch_loss`` first reorders the showers (layers last -> layers first) and adds the
    channel axis (reference experiments/lemurs/model.py:8-104)."""

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
        if not isinstance(unwrap(net), ViT):
            raise TypeError("vit4hep_amd.LEMURSCFM needs a vit4hep_amd.nn.vit.ViT network: the path has no PyTorch fallback")
        self._core()

    def _core(self):
        core = unwrap(self.net)
        core.set_geometry(self.shape, self.patch_shape)
        return core

    def to_patches(self, x):
        return _patchify(self._core(), x.contiguous())

    def from_patches(self, x):
        return _unpatchify(self._core(), x.contiguous())

    def _batch_loss(self, x):
        """reference lemurs/model.py:62-65: (B, R, A, L) -> (B, 1, L, A, R)"""
        x[0] = x[0].permute(0, 3, 2, 1)
        x[0] = x[0].unsqueeze(1)
        return super()._batch_loss(x)

    def forward(self, x, t, c):
        self._core()
        return self.net(x.contiguous(), t, c)

    @torch.inference_mode()
    def sample_batch(self, batch):
        x_T = torch.randn((batch.shape[0], self.in_channels, *self.shape), dtype=batch.dtype, device=batch.device)
        return self._sample_from(x_T, batch)
