"""TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32/fp64) of the reference's ViT-CFM hot path,
used as the checker by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``vit4hep_amd/`` may import
this package: the product path is the HIP library and fails loudly without it.
"""
