"""Warm (same buffers) timing of one contraction at two row counts: fixed cost (prologue) and per-16-row-tile cost of the weight-stationary kernel vs the ring kernel.
usage: python tools/experiments/gemm3_warm.py [qkv|proj|fc1|dproj|dfc2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit4hep_amd import _lib

lib = _lib.load()
dev, dt = "cuda:0", torch.bfloat16
s = _lib.stream_ptr(dev)
for which in sys.argv[1:] or ["qkv"]:
    if which.startswith("J"):
        J, K, qks = int(which[1:]), 480, 0
    else:
        J, K, qks = {"qkv": (1440, 480, 0), "proj": (480, 480, 0), "fc1": (1920, 480, 0), "dproj": (480, 480, 1), "dfc2": (1920, 480, 1)}[which]
    res = {}
    for kern in (2, 3):
        _lib.check(lib.v4h_select_contraction_kernel(kern))
        for BT in (17280, 34560):
            P = torch.randn((BT, K), device=dev).to(dt)
            Q = torch.randn((K, J) if qks else (J, K), device=dev).to(dt)
            bias = torch.randn(J, device=dev)
            out = torch.empty((BT, J), device=dev, dtype=dt)
            args = (_lib.MODES["bf16"], _lib.ptr(P), K, 0, _lib.ptr(Q), Q.stride(0), qks, _lib.ptr(bias), _lib.ptr(out), J, 0, BT, J, K, 1, None, s)
            for _ in range(5):
                _lib.check(lib.v4h_op_gemm(*args))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                _lib.check(lib.v4h_op_gemm(*args))
            e1.record(); torch.cuda.synchronize()
            res[(kern, BT)] = e0.elapsed_time(e1) * 1e3 / 40
    lib.v4h_select_contraction_kernel(0)
    for kern, nm in ((2, "ring"), (3, "w-stat")):
        t1, t2 = res[(kern, 17280)], res[(kern, 34560)]
        print(f"{which:6s} {nm:7s}: {t1:6.1f} us at 17280 rows, {t2:6.1f} at 34560 -> fixed {2 * t1 - t2:5.1f} us, per 17280 rows {t2 - t1:5.1f} us (warm, same buffers)")
