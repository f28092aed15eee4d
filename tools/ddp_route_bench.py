"""The DROP-IN data-parallel route rehearsed on one GPU: `DDP(model.net)` + the unchanged `BaseExperiment._step` (reference
experiments/base_experiment.py:161-167, 555-602) under a 1-rank RCCL group, with DDP's bucket all-reduce replaced - through DDP's own comm-hook interface -
by the stand-in kernel of tools/comm_standin.hip (what a ring all-reduce of 8 ranks occupies and moves: 2 * 7/8 of every bucket, 16 workgroups, paced to a
link rate) on a high-priority communication stream.

What it shows: with ONE autograd node for the network every DDP bucket becomes ready when the whole backward has been enqueued - the collective of all
104 MB is exposed behind the pass; with the per-stage nodes (vit4hep_amd/autograd.py, the default whenever a process group exists) DDP's hooks fire stage by
stage and the buckets are reduced beside the remaining stages.

    python tools/ddp_route_bench.py [--steps 20] [--rounds 2]        # runs every variant in a child process each, interleaved; prints a table
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [  # name, staged nodes, communication
    ("no process group (plain drop-in step)", None, "none"),
    ("DDP, single node, 1-rank RCCL all-reduce", 0, "identity"),
    ("DDP, staged nodes, 1-rank RCCL all-reduce", 1, "identity"),
    ("DDP, single node, stand-in 16 wg @ 300 GB/s", 0, "standin:16:300"),
    ("DDP, staged nodes, stand-in 16 wg @ 300 GB/s", 1, "standin:16:300"),
    ("DDP, single node, stand-in 16 wg @ 150 GB/s", 0, "standin:16:150"),
    ("DDP, staged nodes, stand-in 16 wg @ 150 GB/s", 1, "standin:16:150"),
]


def child(idx, steps, warmup):
    import torch
    import torch.distributed as dist

    import bench
    from tools.comm_interference import standin_lib

    name, staged, comm = VARIANTS[idx]
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if staged is not None:
        os.environ["V4H_STAGED_AUTOGRAD"] = str(staged)
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547", "RANK": "0", "WORLD_SIZE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
    w = bench.WORKLOADS["ds2"]
    model = bench.build_model(w, "bf16", "cuda:0")
    moved = [0]
    if staged is not None:
        from torch.nn.parallel import DistributedDataParallel as DDP

        model.net = DDP(model.net, device_ids=[0], find_unused_parameters=False)  # the reference's call
        if comm.startswith("standin"):
            _, nwg, gbps = comm.split(":")
            nwg, gbps = int(nwg), float(gbps)
            sl = standin_lib()
            factor = 2.0 * 7 / 8
            scratch = torch.empty(int(26_100_000 * 4 * factor) + 4096, dtype=torch.uint8, device=dev)
            cs = torch.cuda.Stream(device=dev, priority=-1)

            def hook(state, bucket):
                buf = bucket.buffer()
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                fut = torch.futures.Future(devices=[dev])
                nbytes = buf.numel() * buf.element_size() // 16 * 16
                move = int(nbytes * factor) // 16 * 16
                with torch.cuda.stream(cs):
                    cs.wait_event(ev)
                    rc = sl.comm_standin_copy(scratch.data_ptr(), buf.data_ptr(), nbytes, move, nwg, 16384, gbps, cs.cuda_stream)
                    assert rc == 0, rc
                    moved[0] += move
                    fut.set_result(buf)
                return fut

            model.net.register_comm_hook(None, hook)
    opt = torch.optim.AdamW([{"params": model.parameters(), "lr": 1e-4}], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50000, eta_min=0)
    x, c = bench.synthetic(w["shape"], w["B"], seed=4, device="cuda:0", cond=w["cond"])
    model.train()

    def ref_step():  # BaseExperiment._step, line by line
        loss = model._batch_loss([x, c])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.net.parameters(), float("inf")).cpu().item()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000.0, error_if_nonfinite=True).cpu().item()
        opt.step()
        sched.step()
        return loss.item(), gnorm

    for _ in range(warmup):
        ref_step()
    torch.cuda.synchronize()
    moved[0] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, gn = ref_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"variant": name, "steps_per_s": steps / dt, "mb_moved_per_step": moved[0] / steps / 1e6, "loss": loss}), flush=True)
    if staged is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", type=int, default=-1)
    args = ap.parse_args()
    if args.child >= 0:
        return child(args.child, args.steps, args.warmup)
    res = {v[0]: [] for v in VARIANTS}
    moved = {}
    for r in range(args.rounds):
        for i, (name, _, _) in enumerate(VARIANTS):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(i), "--steps", str(args.steps), "--warmup", str(args.warmup)], capture_output=True,
                               text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"variant"')]
            if p.returncode != 0 or not line:
                print(f"# {name}: failed (rc {p.returncode}): {p.stderr[-1500:]}", file=sys.stderr)
                continue
            rec = json.loads(line[-1])
            res[name].append(rec["steps_per_s"])
            moved[name] = rec["mb_moved_per_step"]
            print(f"# round {r + 1} {name}: {rec['steps_per_s']:.2f} steps/s", file=sys.stderr, flush=True)
    base = res[VARIANTS[0][0]]
    base = sum(base) / len(base) if base else float("nan")
    print(f"ds2 bs=128 bf16, unchanged BaseExperiment._step through DDP(model.net); {args.steps} timed steps x {args.rounds} interleaved rounds, one process per variant, one box")
    print(f"{'variant':48s} {'steps/s (rounds)':22s} {'mean':>8s} {'vs no group':>12s} {'MB moved/step':>14s}")
    for name, _, _ in VARIANTS:
        if not res[name]:
            continue
        m = sum(res[name]) / len(res[name])
        print(f"{name:48s} {' '.join(f'{v:7.2f}' for v in res[name]):22s} {m:8.2f} {m / base:12.3f} {moved.get(name, 0.0):14.1f}")


if __name__ == "__main__":
    main()
