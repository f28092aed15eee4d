"""Register / scratch summary of the built kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [substring]   (after `python -m vit4hep_amd.build --report`)"""
import glob
import os
import re
import sys

here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit4hep_amd", "_build")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for f in sorted(glob.glob(os.path.join(here, "*.resources.txt"))):
    t = open(f).read()
    for b in re.split(r"remark: Function Name: ", t)[1:]:
        name = b.split(" ")[0]
        if pat not in name:
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        print(f"{name[:110]:110s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} SGPR {g('TotalSGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize .bytes/lane.'):>4s} occ {g('Occupancy .waves/SIMD.')}")
