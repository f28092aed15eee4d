"""Host mirror of the reference's experiments/calogan/model.py: CaloGANCFM."""

from __future__ import annotations

from ...models.base_model import CFM
from ...models.segmented import SegmentedPatching


class CaloGANCFM(SegmentedPatching, CFM):
    """CaloGAN e+/gamma/pi+ showers: 3 layers, per-layer patch shapes (reference experiments/calogan/model.py:8-120)."""

    def __init__(self, net, list_shape, list_edges, list_patch_shape, in_channels=1, time_distribution="uniform", trajectory="linear",
                 odeint_kwargs=None, *args, **kwargs):
        CFM.__init__(self, None, time_distribution, trajectory, odeint_kwargs, *args, **kwargs)
        self.shape = [int(s) for s in self.shape]
        self.in_channels = in_channels
        self.list_patch_shape = [list(int(v) for v in p) for p in list_patch_shape]
        self._init_segments(net, list_shape, list_edges, self.list_patch_shape)
