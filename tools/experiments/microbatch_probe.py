"""Probe: would two half-batches in flight on two streams use the chip better than one full batch on one stream?

The training forward of ds2 (bs 128) is a strict chain of kernels on one stream; three of its four block contractions are 0.8 tile round (204 tiles on
256 CUs) and the LayerNorm kernels are HBM-bound while the contractions are not.  This times, on one GPU:

  A. the training forward of the full batch on one stream (what the step does),
  B. two training forwards of half a batch each, on two streams, enqueued alternately by one host thread (same weights, own workspaces),
  C. one half-batch forward alone (the lower bound of B if the two overlapped perfectly ... and the upper bound x 2 if not at all).

    python tools/experiments/microbatch_probe.py [--reps 30]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from vit4hep_amd.autograd import run_forward  # noqa: E402
from vit4hep_amd.trainer import CFMTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--workload", default="ds2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    w = bench.WORKLOADS[args.workload]
    model = bench.build_model(w, "bf16", "cuda:0")
    tr = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000)
    net, params = tr.net, tr.p_views
    B = w["B"]
    x, c = bench.synthetic(w["shape"], B, seed=1, device="cuda:0", cond=w["cond"])
    t = torch.rand(B, device=dev)
    plan = net._get_plan()
    h = B // 2

    def ws_for(b):
        return torch.empty(plan.workspace_bytes(b, True), dtype=torch.uint8, device=dev)

    ws_full, ws_a, ws_b, = ws_for(B), ws_for(h), ws_for(h)
    xa, xb, ca, cb, ta, tb = x[:h].contiguous(), x[h:].contiguous(), c[:h].contiguous(), c[h:].contiguous(), t[:h].contiguous(), t[h:].contiguous()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def full(reuse):
        return run_forward(net, params, x, t, c, True, ws=ws_full, reuse_operands=reuse)[0]

    def half(which, reuse):
        if which == 0:
            return run_forward(net, params, xa, ta, ca, True, ws=ws_a, reuse_operands=reuse)[0]
        return run_forward(net, params, xb, tb, cb, True, ws=ws_b, reuse_operands=reuse)[0]

    # operand copies into each workspace once; correctness of the split (rows are independent)
    o_full = full(False)
    with torch.cuda.stream(s1):
        o_a = half(0, False)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        o_b = half(1, False)
    torch.cuda.synchronize()
    err = (torch.cat([o_a, o_b]) - o_full).abs().max().item()
    print(f"max |two halves - full| = {err:.3e}")

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        for s in (s1, s2):
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def run_a():
        full(True)

    def run_b():
        s1.wait_stream(torch.cuda.current_stream())
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            half(0, True)
        with torch.cuda.stream(s2):
            half(1, True)

    def run_c():
        with torch.cuda.stream(s1):
            half(0, True)

    for rnd in range(3):
        a = timed(run_a, args.reps)
        b = timed(run_b, args.reps)
        cc = timed(run_c, args.reps)
        print(f"round {rnd}: full batch on one stream {a:8.1f} us | two halves on two streams {b:8.1f} us ({b / a:.3f}x) | one half alone {cc:8.1f} us")


if __name__ == "__main__":
    main()
