"""What does a gradient all-reduce running BESIDE the update step cost it?  Rehearsed on one GPU (no 8-GPU node is available to this build).

Under N ranks every backward stage's gradient slice is all-reduced by RCCL on a communication stream, behind the event the library records when the
slice is final (vit4hep_amd/parallel.py; reference: DDP(model.net), experiments/base_experiment.py:161-167).  RCCL's ring kernel is a few long-lived
workgroups that hold CUs while the xGMI links move the bucket.  On one GPU a 1-rank all-reduce is the identity and occupies nothing, so this tool swaps
the collective for a stand-in kernel that occupies what the real one would (tools/comm_standin.hip: `nwg` workgroups of 256 lanes, `lds` bytes of LDS
each, 16-byte loads of the slice, system-scope stores, paced to a link rate), enqueued exactly where the all-reduces go, and moving what a ring
all-reduce moves through every rank: 2 (N-1)/N of the bucket = 182 MB per step for the 104 MB of f32 gradients at N = 8.

Variants, interleaved on the same box (box-to-box spread is +-4 %, so only same-box differences mean anything):
  single       the single-rank path (no process group): the driver's N = 1 number
  identity     process group + V4H_FORCE_COLLECTIVES=1, RCCL's 1-rank all-reduce: the cost of the stage events / stream plumbing alone
  nwg=K[,reserve=R][,gbps=G]   the stand-in with K workgroups; R = CUs the persistent kernels of the BACKWARD pass leave free (v4h_reserve_compute_units)

usage: python tools/comm_interference.py [--steps 30] [--rounds 2] [--lds 16384] > gpurun_out/comm_interference.txt
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def standin_lib():
    so = os.path.join(ROOT, "tools", "libcomm_standin.so")
    src = os.path.join(ROOT, "tools", "comm_standin.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.comm_standin_copy.restype = ctypes.c_int
    lib.comm_standin_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--lds", type=int, default=16384)
    ap.add_argument("--ranks", type=int, default=8, help="ranks of the rehearsed job: a ring all-reduce moves 2 (ranks-1)/ranks of every bucket")
    ap.add_argument("--grad-allreduce", default=None, choices=[None, "bf16"], help="bf16: CFMTrainer(grad_allreduce='bf16') - the stand-in then moves half the bytes, the casts run for real")
    ap.add_argument("--only", default="", help="comma-free substring filter on variant names, ';'-separated (e.g. 'single;nwg=16,reserve=0,gbps=300')")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda:0"))

    import bench
    from vit4hep_amd import _lib
    from vit4hep_amd.trainer import CFMTrainer

    lib = _lib.load()
    sl = standin_lib()
    w = bench.WORKLOADS["ds2"]
    model = bench.build_model(w, "bf16", "cuda:0")
    trainer = CFMTrainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000, grad_allreduce=args.grad_allreduce)
    x, c = bench.synthetic(w["shape"], w["B"], seed=0, device="cuda:0", cond=w["cond"])
    torch.manual_seed(1000)
    factor = 2.0 * (args.ranks - 1) / args.ranks
    scratch = torch.empty(int(trainer.flat_g.numel() * 4 * factor) + 4096, dtype=torch.uint8, device="cuda:0")
    real_all_reduce = dist.all_reduce
    cur = {"nwg": 0, "gbps": 0.0, "moved": 0}

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        if cur["nwg"] == 0 or t.numel() < 1024:
            return real_all_reduce(t, op=op, group=group, async_op=async_op)
        if cur["nwg"] < 0:  # no collective at all: only the event the main stream will wait for
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return _Work(ev)
        nbytes = t.numel() * t.element_size() // 16 * 16
        move = int(nbytes * factor) // 16 * 16
        s = torch.cuda.current_stream()
        rc = sl.comm_standin_copy(scratch.data_ptr(), t.data_ptr(), nbytes, move, cur["nwg"], args.lds, cur["gbps"], s.cuda_stream)
        assert rc == 0, rc
        cur["moved"] += move
        ev = torch.cuda.Event()
        ev.record(s)
        return _Work(ev)

    dist.all_reduce = fake_all_reduce

    variants = [("single", None), ("identity", dict(nwg=0, reserve=0, gbps=0.0))]
    for nwg in (8, 16):
        for gbps in (150.0, 300.0, 0.0):
            for reserve in (0, (nwg + 7) // 8 * 8):
                variants.append((f"nwg={nwg},reserve={reserve},gbps={gbps:g}", dict(nwg=nwg, reserve=reserve, gbps=gbps)))
    variants.append(("identity,reserve=16", dict(nwg=0, reserve=16, gbps=0.0)))
    variants.append(("events+waits, no collective", dict(nwg=-1, reserve=0, gbps=0.0)))
    variants.append(("events only", dict(nwg=-2, reserve=0, gbps=0.0)))

    if args.only:
        variants = [v for v in variants if v[0] == "single" or any(o == v[0] for o in args.only.split(";"))]

    def comm_alone(cfg):
        """the stand-in's own time for one step's buckets, nothing beside it (ms)"""
        if cfg is None or cfg["nwg"] <= 0:
            return 0.0
        cur["nwg"], cur["gbps"] = cfg["nwg"], cfg["gbps"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            for lo, hi in trainer.stage_slices:
                fake_all_reduce(trainer.flat_g[lo:hi])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3

    real_after = trainer.reducer.reduce_slice_after

    def run(cfg):
        trainer.reducer.reduce_slice_after = (lambda lo, hi, ev: None) if cfg is not None and cfg["nwg"] == -2 else real_after
        if cfg is None:
            os.environ["V4H_FORCE_COLLECTIVES"] = "0"
        else:
            os.environ["V4H_FORCE_COLLECTIVES"] = "1"
            cur["nwg"], cur["gbps"] = cfg["nwg"], cfg["gbps"]
            trainer.comm_reserve_cus = cfg["reserve"]  # (held only while the backward pass is enqueued)
        for _ in range(args.warmup):
            trainer.step(x, c)
        torch.cuda.synchronize()
        cur["moved"] = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, gn = trainer.step(x, c)
        host = time.perf_counter() - t0  # the host is done enqueueing; what is left is the device's backlog
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        CFMTrainer.check_finite(gn)
        host_share[0] = host / dt
        return args.steps / dt, cur["moved"] / args.steps / 1e6

    res = {name: [] for name, _ in variants}
    moved, hostb = {}, {}
    host_share = [0.0]
    alone = {name: comm_alone(cfg) for name, cfg in variants}
    for r in range(args.rounds):
        for name, cfg in variants:
            v, mb = run(cfg)
            res[name].append(v)
            moved[name] = mb
            hostb[name] = host_share[0]
            print(f"# round {r + 1} {name}: {v:.2f} steps/s", file=sys.stderr, flush=True)
    base = sum(res["single"]) / len(res["single"])
    print(f"ds2 bs=128 bf16 update step, {args.steps} timed steps x {args.rounds} interleaved rounds on one box; stand-in LDS {args.lds} B per workgroup; "
          f"rehearsed ring of {args.ranks} ranks")
    print(f"{'variant':34s} {'steps/s (rounds)':28s} {'mean':>8s} {'vs single':>10s} {'MB moved/step':>14s} {'stand-in alone ms/step':>23s} {'host enqueue / wall':>20s}")
    for name, _ in variants:
        m = sum(res[name]) / len(res[name])
        print(f"{name:34s} {' '.join(f'{v:7.2f}' for v in res[name]):28s} {m:8.2f} {m / base:10.3f} {moved[name]:14.1f} {alone[name]:23.2f} {hostb[name]:20.2f}")
    dist.all_reduce = real_all_reduce
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
