"""Autograd plumbing of the ViT: forward and backward both run inside libvit4hep_hip.so.

Single process: ONE autograd node for the whole network (one forward call, one backward call).
Under ``torch.distributed`` (the reference's ``DDP(model.net)``, experiments/base_experiment.py:161-167): the same single forward call, but a CHAIN of
nodes - embedders, block 0 ... block depth-1, final layer - whose backward each runs one stage of the library's staged backward pass
(include/vit4hep_hip.h: stage 0 final layer, 1+j block depth-1-j, depth+1 embedders) and returns only that stage's parameter gradients.  DDP's reducer
hooks therefore fire stage by stage, as they do for the reference's layer-by-layer autograd graph, and its bucketed all-reduce overlaps the rest of the
backward pass instead of starting after it.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


def _on_device_of(argpos):
    """Decorator: run the wrapped call with the device of its tensor argument ``argpos`` current (see _lib.on_device)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kw):
            ref = args[argpos] if len(args) > argpos else next((v for v in list(args) + list(kw.values()) if isinstance(v, torch.Tensor) and v.is_cuda), None)
            if not (isinstance(ref, torch.Tensor) and ref.is_cuda):  # no device tensor to go by: the call itself raises the proper error
                return fn(*args, **kw)
            with _lib.on_device(ref):
                return fn(*args, **kw)

        return wrapped

    return deco


def _vox_shape(net, B):
    return (B, 1, *net.voxel_shape())


@_on_device_of(1)
def _patchify(net, vox):
    plan = net._get_plan()
    B = vox.shape[0]
    if tuple(vox.shape) != _vox_shape(net, B):
        raise RuntimeError(f"to_patches: input shape {tuple(vox.shape)} does not match the geometry {_vox_shape(net, B)}")
    pmap, _ = net.device_tables(vox.device)
    tok = torch.empty((B, net.num_tokens, int(net.patch_dim)), dtype=torch.float32, device=vox.device)
    _lib.check(_lib.load().v4h_op_patchify(plan.handle, _lib.ptr(vox), _lib.ptr(tok), B, _lib.stream_ptr(vox.device), _lib.ptr(pmap)), "v4h_op_patchify")
    return tok


@_on_device_of(1)
def _unpatchify(net, tok):
    plan = net._get_plan()
    B = tok.shape[0]
    if tuple(tok.shape) != (B, net.num_tokens, int(net.patch_dim)):
        raise RuntimeError(f"from_patches: input shape {tuple(tok.shape)} does not match (B, {net.num_tokens}, {int(net.patch_dim)})")
    pmap, _ = net.device_tables(tok.device)
    # voxels no token maps to (index -1 never occurs for the reference's rearrange patterns) stay zero
    vox = (torch.zeros if net.map_has_holes() else torch.empty)(_vox_shape(net, B), dtype=torch.float32, device=tok.device)
    _lib.check(_lib.load().v4h_op_unpatchify(plan.handle, _lib.ptr(tok), _lib.ptr(vox), B, _lib.stream_ptr(tok.device), _lib.ptr(pmap)), "v4h_op_unpatchify")
    return vox


@_on_device_of(2)
def run_forward(net, params, x_vox, t, c, training, ws=None, reuse_operands=False):
    """Enqueue CaloChallengeCFM.forward on voxels.  Returns (out_vox, workspace).  ``reuse_operands`` (training, caller-owned ``ws``): the operand
    copies of exactly these parameter values are already in ``ws`` (``prepare_operands`` after the optimizer update)."""
    plan = net._get_plan()
    B = x_vox.shape[0]
    dev = x_vox.device
    if tuple(x_vox.shape) != _vox_shape(net, B):  # the kernels index by the plan's geometry: never launch on a mismatching buffer
        raise RuntimeError(f"input shape {tuple(x_vox.shape)} does not match the network geometry {_vox_shape(net, B)}")
    flags = 1 if training else 0
    if training and ws is not None and reuse_operands:
        flags |= 2  # V4H_FWD_REUSE_OPERANDS
    if ws is None:
        if training:
            ws = torch.empty(plan.workspace_bytes(B, True), dtype=torch.uint8, device=dev)
        else:  # persistent workspace: frozen weights keep their operand copies between calls (the sampler's 80 evaluations per batch)
            ws = net.inference_workspace(B, dev)
            if net.operands_current(params, ws):
                flags |= 2  # V4H_FWD_REUSE_OPERANDS
                if net.condition_current(c, ws):
                    flags |= 4  # V4H_FWD_SAME_CONDITION: the ODE solver's evaluations of one batch share their conditions
            else:
                net.condition_current(c, ws)  # remember these conditions: their embedding is (re)computed by this call
    net._last_fwd_flags = flags  # introspection (tests)
    out = torch.zeros_like(x_vox) if net.map_has_holes() else torch.empty_like(x_vox)
    tab = _lib.pointer_table(params)
    pmap, pos = net.device_tables(dev)
    _lib.check(
        _lib.load().v4h_vit_forward(plan.handle, B, tab, _lib.ptr(x_vox), _lib.ptr(t), _lib.ptr(c), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                    flags, _lib.stream_ptr(dev), _lib.ptr(pmap), _lib.ptr(pos)),
        "v4h_vit_forward",
    )
    return out, ws


@_on_device_of(2)
def prepare_operands(net, params, ws, B):
    """Operand copies + positional table of ``params`` into the training workspace ``ws`` on the plan's side stream, behind everything enqueued so far
    (include/vit4hep_hip.h: v4h_vit_prepare_operands); the next training forward on ``ws`` with batch ``B`` may pass ``reuse_operands``."""
    plan = net._get_plan()
    _, pos = net.device_tables(ws.device)
    _lib.check(_lib.load().v4h_vit_prepare_operands(plan.handle, int(B), _lib.pointer_table(params), _lib.ptr(ws), ws.numel(), 1, _lib.stream_ptr(ws.device),
                                                    _lib.ptr(pos)), "v4h_vit_prepare_operands")


@_on_device_of(2)
def update_ahead(net, params, flat, offsets, ws, B, gnorm_sq, hyper, st_in, st_out, nonfinite, gnorm_out):
    """The clip + AdamW update of the flat buffers ``flat = (p, g, m, v)`` pipelined into the next step (include/vit4hep_hip.h: v4h_vit_update_ahead):
    staged on the plan's side stream, the next training forward on ``ws`` (``reuse_operands``) waits stage by stage.  ``offsets``: ctypes int64 array."""
    plan = net._get_plan()
    _, pos = net.device_tables(ws.device)
    p, g, m, v = flat
    _lib.check(
        _lib.load().v4h_vit_update_ahead(plan.handle, int(B), _lib.pointer_table(params), _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), offsets, _lib.ptr(ws),
                                         ws.numel(), _lib.ptr(gnorm_sq), *hyper, _lib.ptr(st_in), _lib.ptr(st_out), _lib.ptr(nonfinite), _lib.ptr(gnorm_out),
                                         _lib.stream_ptr(ws.device), _lib.ptr(pos)),
        "v4h_vit_update_ahead",
    )


def plan_join(net, device):
    """The current stream of ``device`` waits for everything the plan's side stream holds (pipelined update, operand copies)."""
    plan = net._plan
    if plan is not None:
        with _lib.on_device(torch.empty(0, device=device)):
            _lib.check(_lib.load().v4h_plan_join(plan.handle, _lib.stream_ptr(device)), "v4h_plan_join")


@_on_device_of(4)
def run_backward(net, params, grads, dout_vox, ws, stage_first=0, stage_last=None, tables=None):
    """``tables``: (parameter pointer table, gradient pointer table) made once by the caller of a staged pass (building two ctypes arrays of 75
    pointers per stage call is a measurable share of a host-bound step)."""
    plan = net._get_plan()
    B = dout_vox.shape[0] if dout_vox is not None else None
    if stage_last is None:
        stage_last = plan.num_stages - 1
    pmap, pos = net.device_tables(ws.device)
    ptab, gtab = tables if tables is not None else (_lib.pointer_table(params), _lib.pointer_table(grads))
    _lib.check(
        _lib.load().v4h_vit_backward(plan.handle, B, ptab, gtab, _lib.ptr(dout_vox), _lib.ptr(ws), ws.numel(),
                                     stage_first, stage_last, _lib.stream_ptr(ws.device), _lib.ptr(pmap), _lib.ptr(pos)),
        "v4h_vit_backward",
    )


@_on_device_of(4)
def run_backward_events(net, params, grads, dout_vox, ws, stage_events):
    """The whole backward pass as one library call; ``stage_events[s]`` (torch.cuda.Event, already created) is recorded as soon as the
    gradients of stage s are final (include/vit4hep_hip.h: v4h_vit_backward_events)."""
    plan = net._get_plan()
    if len(stage_events) != plan.num_stages:
        raise RuntimeError(f"need {plan.num_stages} stage events, got {len(stage_events)}")
    handles = (_lib.C.c_void_p * len(stage_events))()
    for i, ev in enumerate(stage_events):
        h = ev.cuda_event
        if not h:
            raise RuntimeError("stage event not created yet: record it once before the first use")
        handles[i] = h
    pmap, pos = net.device_tables(ws.device)
    _lib.check(
        _lib.load().v4h_vit_backward_events(plan.handle, dout_vox.shape[0], _lib.pointer_table(params), _lib.pointer_table(grads), _lib.ptr(dout_vox), _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr(ws.device), _lib.ptr(pmap), _lib.ptr(pos), handles),
        "v4h_vit_backward_events",
    )


def _prep_inputs(net, x, t, c):
    x = _lib.require_cuda(x, "x")
    c = _lib.require_cuda(c, "c")
    t = _lib.require_cuda(t, "t").reshape(-1)
    B = x.shape[0]
    if t.numel() != B or c.shape != (B, int(net.condition_dim)):
        raise RuntimeError(f"bad conditioning shapes: t {tuple(t.shape)}, c {tuple(c.shape)} for batch {B}")
    return x, t, c


class _ViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, t, c, patches_io, *params):
        x_vox = _unpatchify(net, x) if patches_io else x
        if tuple(x_vox.shape) != _vox_shape(net, x_vox.shape[0]):
            raise RuntimeError(f"input shape {tuple(x.shape)} does not match the network geometry {_vox_shape(net, x_vox.shape[0])}")
        training = any(ctx.needs_input_grad[5:])
        detached = [p.detach() for p in params]
        out, ws = run_forward(net, detached, x_vox, t, c, training)
        if training:
            ctx.net, ctx.ws, ctx.patches_io = net, ws, patches_io
            ctx.save_for_backward(*params)
        return _patchify(net, out) if patches_io else out

    @staticmethod
    def backward(ctx, grad_out):
        net = ctx.net
        params = [p.detach() for p in ctx.saved_tensors]
        g = _lib.require_cuda(grad_out, "grad_output")
        dout = _unpatchify(net, g) if ctx.patches_io else g
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]  # 256-byte aligned slices of one buffer
        # the node owns this buffer: the pass WRITES every gradient (gradient mode 1 around this call only; the plan is shared with accumulating callers),
        # so no 104 MB zero fill per backward
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dout.device)
        grads, off = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[off : off + p.numel()].view_as(p))
            off += n
        plan_h = net._get_plan().handle
        _lib.check(_lib.load().v4h_plan_set_gradient_mode(plan_h, 1), "v4h_plan_set_gradient_mode")
        try:
            run_backward(net, params, grads, dout, ctx.ws)
        finally:
            _lib.load().v4h_plan_set_gradient_mode(plan_h, 0)
        ctx.ws = None
        return (None, None, None, None, None, *grads)


# ---------------------------------------------------------------------------------------------------------------- staged nodes (data-parallel runs)
STAGE_LOG = None  # tests: a list that receives ("stage", s) when backward stage s starts (tests/test_hip_round4.py)


class _Pass:
    """State shared by the nodes of one forward / backward pass."""

    __slots__ = ("net", "ws", "patches_io", "params", "grads", "dout", "out", "carrier_grad", "depth", "tables", "late", "events")


def _stage_param_slices(net, nparams):
    """Positions in parameter_list() owned by each backward stage: [embedders (+ fine-tuning mappers)], [block 0], ..., [block depth-1], [final layer].
    The layout is the library's (csrc/v4h_runtime.hip: 11 embedder tensors, 10 per block, 4 of the final layer, mappers behind): checked against the plan."""
    depth = int(net.depth)
    plan = net._get_plan()
    if plan.num_stages != depth + 2 or nparams != plan.num_params or nparams < 11 + 10 * depth + 4:
        raise RuntimeError(f"staged autograd: parameter inventory ({nparams} tensors, depth {depth}) does not match the library's ({plan.num_params} tensors, "
                           f"{plan.num_stages} backward stages)")
    first = list(range(0, 11)) + list(range(11 + 10 * depth + 4, nparams))
    blocks = [list(range(11 + 10 * i, 11 + 10 * (i + 1))) for i in range(depth)]
    final = list(range(11 + 10 * depth, 11 + 10 * depth + 4))
    if [plan.shapes[k][0] for k in (final[0], final[2])] != [int(net.final_layer.linear.weight.shape[0]), 2 * int(net.hidden_dim)]:
        raise RuntimeError("staged autograd: the final layer's tensors are not where the stage map expects them")
    return first, blocks, final


def _stage_events(net, n, device):
    """One torch.cuda.Event per backward stage, created once per network (the library records them: v4h_vit_backward_stage)."""
    evs = getattr(net, "_stage_evs", None)
    if evs is None or len(evs) != n or evs[0][1] != torch.device(device):
        evs = []
        for _ in range(n):
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(device))  # torch creates the hipEvent_t at the first record
            evs.append((e, torch.device(device)))
        net._stage_evs = evs
    return [e for e, _ in evs]


def _run_stage(ps, stage):
    """One backward stage.  Unshifted chain (some parameters frozen): the call ends with a join of the library's two streams, the stage's gradients are
    final in stream order when it returns.  Shifted chain (`ps.late`, every parameter trainable - the data-parallel training case): no join, the library
    records the stage's event, and the gradients are handed to autograd by the NEXT node behind a wait for that event (`_await_stage`)."""
    if STAGE_LOG is not None:
        STAGE_LOG.append(("stage", stage))
    if ps.tables is None:
        ps.tables = (_lib.pointer_table(ps.params), _lib.pointer_table(ps.grads))
    plan = ps.net._get_plan()
    lib = _lib.load()
    # The pass owns its private gradient buffer: every stage WRITES its gradients (gradient mode 1, set around this call only - the plan is shared with
    # accumulating callers), so the buffer is allocated without the 104 MB zero fill.
    _lib.check(lib.v4h_plan_set_gradient_mode(plan.handle, 1), "v4h_plan_set_gradient_mode")
    try:
        if not ps.late:
            run_backward(ps.net, ps.params, ps.grads, ps.dout, ps.ws, stage, stage, tables=ps.tables)
            return
        dev = ps.ws.device
        if ps.events is None:
            ps.events = _stage_events(ps.net, plan.num_stages, dev)
        pmap, pos = ps.net.device_tables(dev)
        last = stage == plan.num_stages - 1
        with _lib.on_device(ps.ws):
            _lib.check(
                lib.v4h_vit_backward_stage(plan.handle, ps.dout.shape[0], ps.tables[0], ps.tables[1], _lib.ptr(ps.dout), _lib.ptr(ps.ws), ps.ws.numel(), stage,
                                           _lib.stream_ptr(dev), _lib.ptr(pmap), _lib.ptr(pos), None if last else ps.events[stage].cuda_event, 1 if last else 0),
                "v4h_vit_backward_stage",
            )
    finally:
        lib.v4h_plan_set_gradient_mode(plan.handle, 0)


def _await_stage(ps, stage):
    """The current stream waits for the event of an EARLIER stage (long reached: one wait packet, no stall)."""
    torch.cuda.current_stream(ps.ws.device).wait_event(ps.events[stage])


class _StageEmbed(torch.autograd.Function):
    """Runs the WHOLE forward (one library call); its backward is the last stage (x / t / c embedders, positional frequencies, mappers)."""

    @staticmethod
    def forward(ctx, ps, x_vox, t, c, idx, *stage_params):
        out, ps.ws = run_forward(ps.net, ps.params, x_vox, t, c, True)
        ps.out = out
        ctx.ps, ctx.idx = ps, idx
        return torch.empty(1, dtype=torch.float32, device=x_vox.device)  # carrier: links the nodes, carries no data

    @staticmethod
    def backward(ctx, _carrier_grad):
        ps = ctx.ps
        _run_stage(ps, ps.depth + 1)  # (ends with the full join: block 0's gradients, handed over here on the shifted chain, are final too)
        grads = [ps.grads[i] for i in ctx.idx]
        ps.ws = ps.dout = None
        return (None, None, None, None, None, *grads)


class _StageBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ps, carrier, i, idx, *stage_params):
        ctx.ps, ctx.i, ctx.idx = ps, i, idx
        return torch.empty_like(carrier)

    @staticmethod
    def backward(ctx, _carrier_grad):
        ps = ctx.ps
        st = 1 + (ps.depth - 1 - ctx.i)
        _run_stage(ps, st)
        if ps.late:  # ctx.idx are the parameters of the stage BEFORE this one: their event was recorded a whole stage ago
            _await_stage(ps, st - 1)
        return (None, ps.carrier_grad, None, None, *[ps.grads[k] for k in ctx.idx])


class _StageFinal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ps, carrier, idx, *stage_params):
        ctx.ps, ctx.idx = ps, idx
        out, ps.out = ps.out, None
        return _patchify(ps.net, out) if ps.patches_io else out

    @staticmethod
    def backward(ctx, grad_out):
        ps = ctx.ps
        g = _lib.require_cuda(grad_out, "grad_output")
        ps.dout = _unpatchify(ps.net, g) if ps.patches_io else g
        sizes = [(p.numel() + 63) // 64 * 64 for p in ps.params]  # 256-byte aligned slices of one buffer (written, not accumulated into: _run_stage)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=ps.dout.device)
        ps.grads, off = [], 0
        for p, n in zip(ps.params, sizes):
            ps.grads.append(flat[off : off + p.numel()].view_as(p))
            off += n
        ps.carrier_grad = torch.zeros(1, dtype=torch.float32, device=ps.dout.device)  # any defined tensor: the carriers carry no data
        _run_stage(ps, 0)
        return (None, ps.carrier_grad, None, *[ps.grads[k] for k in ctx.idx])


def staged_autograd():
    """Per-stage nodes exactly when a process group exists (V4H_STAGED_AUTOGRAD=0 / 1 overrides: A/B and tests)."""
    e = os.environ.get("V4H_STAGED_AUTOGRAD")
    if e in ("0", "1"):
        return e == "1"
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _apply_staged(net, x, t, c, patches_io, params):
    x_vox = _unpatchify(net, x) if patches_io else x
    if tuple(x_vox.shape) != _vox_shape(net, x_vox.shape[0]):
        raise RuntimeError(f"input shape {tuple(x.shape)} does not match the network geometry {_vox_shape(net, x_vox.shape[0])}")
    ps = _Pass()
    ps.net, ps.patches_io, ps.params, ps.depth = net, patches_io, [p.detach() for p in params], int(net.depth)
    ps.ws = ps.grads = ps.dout = ps.out = ps.carrier_grad = ps.tables = ps.events = None
    first, blocks, final = _stage_param_slices(net, len(params))
    # Shifted hand-over (round 5): with every parameter trainable each node returns the gradients of the stage that ran BEFORE it in the backward pass -
    # block i's node those of block i + 1 (block depth-1's those of the final layer), the embedder node its own and block 0's -, so that no stage has to
    # join the library's weight-gradient stream before it returns: the gradients it hands on were finished a whole stage earlier (an event wait that is long
    # satisfied).  DDP sees every bucket one stage later and the main stream never stalls for the side stream's tail: 142 -> see profiles/r05_notes.md.
    ps.late = all(p.requires_grad for p in params) and os.environ.get("V4H_STAGED_LATE", "1") != "0"
    if ps.late:
        depth = len(blocks)
        owned_embed = first + (blocks[0] if depth else final)
        carrier = _StageEmbed.apply(ps, x_vox, t, c, owned_embed, *[params[i] for i in owned_embed])
        for i in range(depth):
            idx = blocks[i + 1] if i + 1 < depth else final
            carrier = _StageBlock.apply(ps, carrier, i, idx, *[params[k] for k in idx])
        return _StageFinal.apply(ps, carrier, [])
    carrier = _StageEmbed.apply(ps, x_vox, t, c, first, *[params[i] for i in first])
    for i, idx in enumerate(blocks):
        carrier = _StageBlock.apply(ps, carrier, i, idx, *[params[k] for k in idx])
    if not carrier.requires_grad:  # every parameter below the final layer is frozen: the chain needs no carrier
        carrier = carrier.detach()
    return _StageFinal.apply(ps, carrier, final, *[params[k] for k in final])


def vit_apply(net, x, t, c, patches_io):
    x, t, c = _prep_inputs(net, x, t, c)
    params = net.parameter_list()
    for p in params:
        if not p.is_cuda or p.dtype != torch.float32:
            raise RuntimeError("vit4hep_amd: parameters must be float32 tensors on the MI355X device (model.to(device, torch.float32))")
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if staged_autograd():
            return _apply_staged(net, x, t, c, patches_io, params)
        return _ViTFunction.apply(net, x, t, c, patches_io, *params)
    x_vox = _unpatchify(net, x) if patches_io else x
    if tuple(x_vox.shape) != _vox_shape(net, x_vox.shape[0]):
        raise RuntimeError(f"input shape {tuple(x.shape)} does not match the network geometry {_vox_shape(net, x_vox.shape[0])}")
    out, _ = run_forward(net, [p.detach() for p in params], x_vox, t, c, False)
    return _patchify(net, out) if patches_io else out
