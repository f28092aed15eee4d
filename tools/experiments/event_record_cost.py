"""What an event record between two kernels of one stream costs the stream (the backward pass records one behind every kernel whose output a weight gradient on the
side stream reads): N back-to-back launches of a ~20 us kernel, plain / with a record behind each / with a record + a second stream waiting on it."""
import time, torch
dev = "cuda:0"
x = torch.randn(64 * 1024 * 1024, device=dev)
y = torch.empty_like(x)
side = torch.cuda.Stream()
N = 200
evs = [torch.cuda.Event() for _ in range(N)]
def run(mode):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(N):
        torch.mul(x, 1.0001, out=y)
        if mode >= 1:
            evs[i].record()
        if mode >= 2:
            side.wait_event(evs[i])
            with torch.cuda.stream(side):
                y[:1024].add_(1.0)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N
for mode, name in ((0, "plain"), (1, "record behind every kernel"), (2, "record + side stream waits and runs a tiny kernel")):
    run(mode)
    print(f"{name}: {min(run(mode) for _ in range(3)):.2f} us per kernel")
