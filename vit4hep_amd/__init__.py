"""vit4hep_amd - MI355X-native (gfx950) ViT-CFM hot path of luigifvr/vit4hep.

    from vit4hep_amd import ViT, CFM, CaloChallengeCFM, CaloChallengeCFM_DS1, CaloGANCFM, CaloHadCFM, LEMURSCFM, ParallelTransformer, ShapeChain
    vit4hep_amd.dropin.install()      # make the reference's dotted paths (nn.vit.ViT, ...) resolve to this package

The device work lives in libvit4hep_hip.so (vit4hep_amd/csrc, C ABI in include/vit4hep_hip.h); build it with
`python -m vit4hep_amd.build`.
"""

__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not need torch.cuda or the built library
    if name == "ViT":
        from .nn.vit import ViT

        return ViT
    if name == "CFM":
        from .models.base_model import CFM

        return CFM
    lazy = {
        "CaloChallengeCFM": ".experiments.calochallenge.calochallenge_cfm.model",
        "CaloChallengeCFM_DS1": ".experiments.calochallenge.calochallenge_cfm.model",
        "CaloGANCFM": ".experiments.calogan.model",
        "CaloHadCFM": ".experiments.calohadronic.model",
        "LEMURSCFM": ".experiments.lemurs.model",
        "ParallelTransformer": ".nn.cfm.transformer_cfm",   # energy-model network, forward only
        "ShapeChain": ".transforms",                        # fused pre-/post-processing chain
        "CFMTrainer": ".trainer",
    }
    if name in lazy:
        import importlib

        return getattr(importlib.import_module(lazy[name], __name__), name)
    raise AttributeError(name)
