"""Build libvit4hep_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU).

    python -m vit4hep_amd.build [--force] [--report]

One translation unit per .hip file, compiled in parallel, linked into vit4hep_amd/libvit4hep_hip.so.
The .so travels to the GPU box with the repository snapshot (git-ignored, not gpurun-ignored).
"""

from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# V4H_BUILD_TAG=<tag>: a second build beside the product library (objects in _build_<tag>/, library libvit4hep_hip_<tag>.so; load it with
# VIT4HEP_AMD_LIB=...).  The ablation / tuning build the tools under tools/ need:  V4H_BUILD_TAG=abl V4H_EXTRA_FLAGS=-DV4H_ABLATIONS python -m vit4hep_amd.build
TAG = os.environ.get("V4H_BUILD_TAG", "")
OBJ = os.path.join(HERE, "_build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libvit4hep_hip" + ("_" + TAG if TAG else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"] + os.environ.get("V4H_EXTRA_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/vit4hep_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def kernel_digest():
    """Digest of what determines the device code: every file under csrc/ and the compiler flags (not the C header's comments).  profiles/step_hbm_traffic.json is
    stamped with it, and bench.py reports a measured `traffic` only for the build it was measured on."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, report):
    out = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", out]
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
    if report:
        with open(os.path.join(OBJ, src + ".resources.txt"), "w") as fh:
            fh.write(r.stderr)
    return out


def build(force=False, report=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "digest")
    dig = _digest()
    if not force and not report and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; libvit4hep_hip.so cannot be built")
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, report), _sources()))
    r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, report="--report" in sys.argv)
