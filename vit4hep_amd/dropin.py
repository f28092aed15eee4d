"""Make the reference's dotted module paths resolve to this package, so its Hydra YAMLs work unchanged:

    _target_: experiments.calochallenge.calochallenge_cfm.model.CaloChallengeCFM     (configs/model/cfm/cfm_ds2_electrons.yaml:1)
    _target_: nn.vit.ViT                                                               (configs/model/cfm/cfm_ds2_electrons.yaml:14)

Call ``vit4hep_amd.dropin.install()`` before ``hydra.utils.instantiate(cfg.model)`` (reference
experiments/base_experiment.py:116), e.g. at the top of main.py.  Only the hot-path modules (network, CFM base, trajectories and the CFM wrappers of each dataset) are aliased; every other
reference module (experiments.base_experiment, datasets, transforms, ...) keeps resolving to the reference's own files.
"""

from __future__ import annotations

import importlib
import sys
import types

ALIASES = {
    "nn.vit": "vit4hep_amd.nn.vit",
    "models.base_model": "vit4hep_amd.models.base_model",
    "models.trajectories": "vit4hep_amd.models.trajectories",
    "experiments.calochallenge.calochallenge_cfm.model": "vit4hep_amd.experiments.calochallenge.calochallenge_cfm.model",
    "experiments.calogan.model": "vit4hep_amd.experiments.calogan.model",
    "experiments.calohadronic.model": "vit4hep_amd.experiments.calohadronic.model",
    "experiments.lemurs.model": "vit4hep_amd.experiments.lemurs.model",
}
# Opt-in: the energy-model network runs FORWARD ONLY in the HIP library (it is sampled, never trained, inside a shape-model run:
# experiments/calochallenge/experiment.py:225-247, 323-342).  Alias it in processes that only sample; a process that trains an
# energy model (`model_type: energy`) must keep the reference's own module.
ENERGY_ALIASES = {"nn.cfm.transformer_cfm": "vit4hep_amd.nn.cfm.transformer_cfm"}
_installed = []


def install(energy_sampler=False):
    for ref_name, ours in {**ALIASES, **(ENERGY_ALIASES if energy_sampler else {})}.items():
        parts = ref_name.split(".")
        for i in range(1, len(parts)):  # parent packages: keep the reference's if importable, else a namespace stub
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                try:
                    importlib.import_module(pkg)
                except Exception:
                    stub = types.ModuleType(pkg)
                    stub.__path__ = []
                    sys.modules[pkg] = stub
                    _installed.append(pkg)
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        parent = sys.modules.get(".".join(parts[:-1]))
        if parent is not None:
            setattr(parent, parts[-1], mod)
        _installed.append(ref_name)


def uninstall():
    while _installed:
        sys.modules.pop(_installed.pop(), None)
