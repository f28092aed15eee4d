"""The tuning tools select ablation / tuning builds of the contraction kernels through v4h_debug_set_gemm_cfg, which exists only in a library compiled
with -DV4H_ABLATIONS (the product library contains exact kernels only):

    V4H_BUILD_TAG=abl V4H_EXTRA_FLAGS=-DV4H_ABLATIONS python -m vit4hep_amd.build
    VIT4HEP_AMD_LIB=$PWD/vit4hep_amd/libvit4hep_hip_abl.so python tools/<tool>.py
"""
import sys


def require_ablation_lib(lib):
    if not hasattr(lib, "v4h_debug_set_gemm_cfg"):
        sys.exit(__doc__)
    return lib
