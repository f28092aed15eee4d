"""Known-good reference for the contraction shapes of one DiT block: torch.matmul (hipBLASLt / rocBLAS) in bf16 on the
same random operands, HIP-event timed.  Not part of the product - a yardstick for tools/gemm_bench.py (methodology: never
infer a ceiling from one's own kernel alone).
usage (GPU box): python tools/blaslt_ref.py [BT]"""
import sys
import torch

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 17280
dev = "cuda:0"
D, M = 480, 1920


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(name, I, J, K, kind):
    if kind == "fwd":      # y = x W^T
        a = torch.randn(I, K, device=dev).bfloat16(); w = torch.randn(J, K, device=dev).bfloat16()
        fn = lambda: torch.matmul(a, w.t())
    elif kind == "dgrad":  # dx = dy W
        a = torch.randn(I, K, device=dev).bfloat16(); w = torch.randn(K, J, device=dev).bfloat16()
        fn = lambda: torch.matmul(a, w)
    else:                  # dW = dy^T x   (I = out features, J = in features, K = tokens)
        a = torch.randn(K, I, device=dev).bfloat16(); w = torch.randn(K, J, device=dev).bfloat16()
        fn = lambda: torch.matmul(a.t(), w)
    us = timeit(fn)
    print(f"{name:34s} I={I:6d} J={J:5d} K={K:6d}  {us:8.1f} us  {2.0*I*J*K/us/1e6:7.1f} TFLOP/s", flush=True)


print("--- torch.matmul bf16 (library GEMM), forward / dgrad")
for nm, J, K in (("fwd qkv", 3 * D, D), ("fwd proj", D, D), ("fwd fc1", M, D), ("fwd fc2", D, M)):
    run(nm, BT, J, K, "fwd")
for nm, J, K in (("dgrad qkv", D, 3 * D), ("dgrad proj", D, D), ("dgrad fc1", D, M), ("dgrad fc2", M, D)):
    run(nm, BT, J, K, "dgrad")
print("--- wgrad")
for nm, I, J in (("wgrad qkv", 3 * D, D), ("wgrad proj", D, D), ("wgrad fc1", M, D), ("wgrad fc2", D, M)):
    run(nm, I, J, BT, "wgrad")
print("--- large square, for scale")
run("square 8192", 8192, 8192, 8192, "fwd")
run("square 4096", 4096, 4096, 4096, "fwd")
