"""Native CFM update step: the semantics of BaseExperiment._step (reference experiments/base_experiment.py:555-602
with configs/training/default.yaml) on flat HBM buffers, without autograd and without per-step host syncs.

  loss  = CFM._batch_loss                         (models/base_model.py:203-218)   -> HIP forward + fused MSE
  grads = loss.backward()                         (:560)                           -> HIP backward, one call (stage events for the buckets)
  DDP gradient averaging                          (:161-167)                       -> bucketed RCCL all-reduce, overlapped
  clip_grad_norm_(..., clip, error_if_nonfinite)  (:573-585)                       -> one norm kernel; clip folded into AdamW
  AdamW(lr, betas, eps, wd) + CosineAnnealingLR   (:592-597, :329-431)             -> one fused kernel over all parameters
  all_reduce(loss, AVG)                           (:600-601)

Loss and gradient norm come back as 0-dim device tensors; call ``check_finite`` (or .item()) when the host needs them.

EMA (``ema_decay=...``, the reference's ``ema: true`` + ``training.ema_decay``, base_experiment.py:127-134,593-594): the shadow parameters are updated inside
the AdamW pass (one more f32 stream); ``ema_state_dict()`` is ``torch_ema.ExponentialMovingAverage.state_dict()``'s layout and ``average_parameters()`` its
context manager (validation under the averaged weights, :630-631).  Optimizers other than AdamW and schedulers other than CosineAnnealingLR
(base_experiment.py:329-431) are not this class's fused update: they raise and point at the autograd route, where any torch optimizer works.

Checkpoints: ``state_dict()`` / ``load_state_dict()`` speak the layout of the reference's checkpoint file (``{"model", "optimizer", "scheduler", "ema"}``,
base_experiment.py:661-677): the "optimizer" entry is a ``torch.optim.AdamW.state_dict()`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in
``model.parameters()`` order, one param group), the "scheduler" entry a ``CosineAnnealingLR.state_dict()`` - so a run of this trainer can be continued by the
reference's ``_init_optimizer`` / ``_init_scheduler`` warm start (:374-388, :420-431) and the other way round.
"""

from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from . import _lib
from .autograd import plan_join, prepare_operands, run_backward, run_backward_events, run_forward, update_ahead
from .parallel import BucketReducer, collectives_enabled, world


def _aligned(n, a=64):
    return (n + a - 1) // a * a


class CFMTrainer:
    NONFINITE_MSG = "The total norm for gradients is non-finite, so it cannot be clipped."  # torch.nn.utils.clip_grad_norm_'s own text

    MIN_STEP_SKIP = 1000  # base_experiment.py:31

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, clip_grad_norm=1000.0, iterations=50000, group=None,
                 nonfinite_check_every=50, max_grad_norm=None, clip_grad_value=None, eta_min=0.0, pipeline_update=None, ema_decay=None,
                 optimizer="AdamW", scheduler="CosineAnnealingLR", grad_allreduce=None):
        from .experiments.calochallenge.calochallenge_cfm.model import _unwrap

        # training.optimizer / training.scheduler of the reference (base_experiment.py:329-431: Adam, AdamW, RAdam, Lion, ScheduleFree; OneCycleLR,
        # CosineAnnealingLR, ReduceLROnPlateau, ...).  The fused update IS AdamW + CosineAnnealingLR (configs/training/default.yaml); anything else trains
        # through the autograd node (vit4hep_amd.dropin + the reference's own _init_optimizer / _init_scheduler), where every torch optimizer works.
        if optimizer != "AdamW":
            raise NotImplementedError(f"CFMTrainer: optimizer {optimizer!r} is not the fused update (AdamW only); use the autograd route "
                                      "(model.net through vit4hep_amd.dropin and the reference's _init_optimizer, experiments/base_experiment.py:329-372)")
        if scheduler != "CosineAnnealingLR":
            raise NotImplementedError(f"CFMTrainer: scheduler {scheduler!r} is not the fused update (CosineAnnealingLR only); use the autograd route "
                                      "(the reference's _init_scheduler, experiments/base_experiment.py:390-431)")

        self.model = model
        self.net = model._core() if hasattr(model, "_core") else _unwrap(model.net)  # _core() (re)binds the wrapper's geometry to the net
        if self.net.x_embed_in() or self.net.c_embed_in():
            raise NotImplementedError("CFMTrainer: networks with a fine-tuning embedding mapper train through the autograd node and a torch optimizer "
                                      "(per-module learning rates, experiment_finetuning.py:173-205)")
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.clip = float(clip_grad_norm) if clip_grad_norm is not None else None
        if clip_grad_value is not None:  # training.clip_grad_value (base_experiment.py:567-572; null in every shipped config, "this is dangerous!")
            raise NotImplementedError("CFMTrainer: clip_grad_value is not built (configs/training/default.yaml: null); use the autograd node with "
                                      "torch.nn.utils.clip_grad_value_ if a run needs it")
        # training.max_grad_norm (base_experiment.py:586-591): after MIN_STEP_SKIP iterations an update whose (pre-clip) gradient norm exceeds it is
        # skipped - optimizer AND scheduler stay where they are.  Decided on the device, no host sync.
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm is not None else None
        self.eta_min = float(eta_min)
        self.iterations = int(iterations)
        self.step_count = 0  # host's count of applied updates; the device's own counters (self._state) are what the arithmetic uses
        self.iteration = 0   # `step` of the reference's training loop: every call of step(), applied or skipped
        self.group = group
        # Pipelined update (opt-in; V4H_PIPELINE_UPDATE=1 or pipeline_update=True): AdamW + the operand copies of the new weights run on the library's
        # side stream, stage by stage in the order the next forward needs them, BESIDE the next step's head and first blocks (the same arithmetic, bit
        # for bit; include/vit4hep_hip.h: v4h_vit_update_ahead).  step() then returns before the update has been ordered into the current stream:
        # everything of this class that reads parameters or moments joins first (finish()), and so must a caller that touches them with its own ops
        # between two steps.  Single rank, no CUDA-graph capture, no embedding mappers.
        self.pipeline_update = (os.environ.get("V4H_PIPELINE_UPDATE") == "1") if pipeline_update is None else bool(pipeline_update)
        self._ahead = None  # (workspace pointer, batch) the operand copies of the update in flight were made for
        # EMA of the parameters (torch_ema.ExponentialMovingAverage(model.parameters(), decay): the shadow starts as a copy of the parameters)
        if ema_decay is not None and not (0.0 <= float(ema_decay) <= 1.0):
            raise ValueError("Decay must be between 0 and 1")  # torch_ema's text
        self.ema_decay = float(ema_decay) if ema_decay is not None else None
        self.flat_ema = None
        self._ema_collected = None
        if self.ema_decay is not None and self.pipeline_update:
            raise NotImplementedError("CFMTrainer: ema_decay with pipeline_update is not built (the pipelined update kernel has no shadow stream)")
        if self.pipeline_update and os.environ.get("V4H_STEP_GRAPH") == "1":
            raise ValueError("CFMTrainer: pipeline_update and V4H_STEP_GRAPH are mutually exclusive (a captured step replays the in-line update, which does "
                             "not refresh the operand copies a pipelined update prepares)")
        # The reference raises on a non-finite gradient norm BEFORE optimizer.step(), also without clipping (max_norm = inf; base_experiment.py:573-585).
        # Here the update kernel skips such a step on the device and bumps a sticky counter - and while that counter is non-zero it skips (and counts)
        # EVERY later update too, so no update is ever applied with a shifted Adam / LR-schedule index.  The host looks at the counter every
        # `nonfinite_check_every` steps (one 4-byte read) and in raise_if_nonfinite(), raises the same error and rewinds step_count by the skipped
        # updates: parameters, moments and step index are then exactly those of the last finite step, as after the reference's raise.  (All ranks see
        # the same all-reduced gradient norm, hence the same counter.)
        self.nonfinite_check_every = int(nonfinite_check_every)
        # gradient all-reduce of the data-parallel path: f32 (default) or "bf16" (half the bytes over xGMI, f32 master gradients; parallel.BucketReducer)
        self.grad_allreduce = grad_allreduce if grad_allreduce is not None else os.environ.get("V4H_GRAD_ALLREDUCE")
        self._flatten()

    # ---------------------------------------------------------------------------------------------- flat buffers
    def _flatten(self):
        params = self.net.parameter_list()
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("CFMTrainer: move the model to the MI355X first (model.to('cuda')); there is no CPU path")
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += _aligned(p.numel())
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.p_views, self.g_views = [], []
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.flat_p[o : o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.p_views.append(view)
                self.g_views.append(self.flat_g[o : o + p.numel()].view_as(p))
        self.params = params
        if getattr(self, "ema_decay", None) is not None:
            old = getattr(self, "flat_ema", None)
            self.flat_ema = old.to(dev) if old is not None and old.numel() == off else self.flat_p.clone()
        # contiguous gradient slice of each backward stage (include/vit4hep_hip.h: stage 0 final layer, 1+j block depth-1-j, last embedders)
        depth = int(self.net.depth)
        blk0 = 11
        bounds = lambda i0, i1: (self.offsets[i0], self.offsets[i1] if i1 < len(params) else self.total)
        self.stage_slices = [bounds(blk0 + 10 * depth, len(params))]
        for j in range(depth):
            i = depth - 1 - j
            self.stage_slices.append(bounds(blk0 + 10 * i, blk0 + 10 * (i + 1)))
        self.stage_slices.append(bounds(0, blk0))
        self._offsets_c = (_lib.C.c_int64 * (len(params) + 1))(*self.offsets, self.total)
        self._last_B = None
        self.reducer = BucketReducer(self.flat_g, self.group, compress=getattr(self, "grad_allreduce", None))
        self.comm_reserve_cus = int(os.environ.get("V4H_COMM_RESERVE_CUS", "0"))  # multiple of 8 in [0, 64]
        self.stage_events = None
        self._scal = torch.zeros(4, dtype=torch.float32, device=dev)  # loss | squared gradient norm | gradient norm of the step in progress
        self.loss, self.gnorm_sq, self.gnorm = self._scal[0], self._scal[1], self._scal[2]
        self.nonfinite = torch.zeros((), dtype=torch.int32, device=dev)
        self._t_ring, self._t_i = None, 0
        # V4H_ASYNC_T=1: t through a ring of pinned buffers (_stage_t) instead of the reference's pageable .to(device).  Measured and left OFF: the
        # steady-state step is the same (4.19 ms either way - the device, not the host, sets the pace), and once the host runs several steps ahead of
        # the device for the first time the runtime stalls ONCE for 18-35 ms (growing its signal / kernel-argument pools) - at an unpredictable step,
        # i.e. possibly inside a short timed region (profiles/r04_notes.md).  The pageable copy is the natural throttle that keeps the host one step ahead.
        self.async_t = os.environ.get("V4H_ASYNC_T", "0") == "1"
        self.use_graph = os.environ.get("V4H_STEP_GRAPH") == "1"
        self._graph, self._graph_warm, self._in_capture = None, 0, False
        self._ws = None  # one training workspace kept across steps (2.6 GB at ds2 bs 128)
        # V4H_PREPARE_AHEAD=1: the operand copies of the weights (bf16 casts of 26 M parameters, 30 us) are requested on the library's side stream at the very
        # start of a step and the forward waits for them only in front of its first weight-consuming kernel, so that they run beside the step's head (noise,
        # trajectory, patch gather).  Measured neutral over four same-box A/Bs (237.3 vs 237.5 steps/s on average): off, the casts stay inside the forward.
        self.prepare_ahead = os.environ.get("V4H_PREPARE_AHEAD", "0") == "1"
        # V4H_OVERWRITE_GRADS=0: A/B hook - zero the gradient buffer every step and let the backward accumulate (the form of rounds 1-3)
        self.overwrite_grads = os.environ.get("V4H_OVERWRITE_GRADS", "1") != "0"
        # The host enqueues a step in a quarter of the time the device needs for it, and nothing in step() waits for the device: an update loop that never
        # reads a result runs dozens of steps (thousands of launches) ahead.  The HIP runtime then grows its signal / kernel-argument pools ONCE, at an
        # unpredictable early step, and the device sits idle for 30-60 ms meanwhile (measured: step 0 ... 4 of a timed region taking 27-62 ms instead of
        # 4.2; profiles/r04_notes.md).  The host therefore stays at most `max_steps_ahead` steps ahead (one event per step, waited for that many steps
        # later): the same steady-state rate, no stall.  0 = unbounded (V4H_MAX_STEPS_AHEAD).
        self.max_steps_ahead = int(os.environ.get("V4H_MAX_STEPS_AHEAD", "2"))
        self._ahead_events = []
        # [applied optimizer steps, scheduler steps, updates skipped for max_grad_norm, -] on the device, two copies used alternately (the update kernel
        # reads one and writes the other: include/vit4hep_hip.h, v4h_adamw_step_sched)
        st = getattr(self, "_state", None)
        cur = st[self._cur].to(dev) if st is not None else torch.zeros(4, dtype=torch.int32, device=dev)
        self._state = [cur.clone(), cur.clone()]
        self._cur = 0

    def _check_alias(self):
        if self._ahead is not None and any(p.data_ptr() != v.data_ptr() for p, v in ((self.params[0], self.p_views[0]), (self.params[-1], self.p_views[-1]))):
            self.finish()
        if any(p.data_ptr() != v.data_ptr() for p, v in ((self.params[0], self.p_views[0]), (self.params[-1], self.p_views[-1]))):
            m, v = self.flat_m, self.flat_v
            self._flatten()  # parameters were re-allocated (e.g. model.to(...)): adopt the new storage, keep the optimizer state (and the counters)
            if m.numel() == self.flat_m.numel() and m.device == self.flat_m.device:
                self.flat_m, self.flat_v = m, v

    T_RING = 4

    def _stage_t(self, t_host, dev):
        """Host-sampled times to the device WITHOUT stopping the host: a `.to(device)` from pageable memory is a synchronous copy, i.e. the host waits for
        the previous step's last kernel at the top of every step and then issues the step's first dozen small launches one launch latency apart while
        the GPU idles between them (the head of the step measured 130 us for 80 us of kernels).  Through a small ring of pinned buffers the copy is truly
        asynchronous and the host runs ahead of the device (by at most T_RING steps: a buffer is reused only after its copy has completed)."""
        n = t_host.numel()
        if self._t_ring is None or self._t_ring[0][0].numel() < n:
            self._t_ring = [[torch.empty(n, dtype=torch.float32).pin_memory(), None] for _ in range(self.T_RING)]
        slot = self._t_ring[self._t_i % self.T_RING]
        self._t_i += 1
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:n].copy_(t_host.reshape(-1).to(torch.float32))
        t = torch.empty(n, dtype=torch.float32, device=dev)
        t.copy_(slot[0][:n], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return t

    def lr_at(self, k):
        """CosineAnnealingLR(T_max=iterations, eta_min=0) after k scheduler steps (closed form)."""
        return self.lr * 0.5 * (1.0 + math.cos(math.pi * k / self.iterations))

    # ---------------------------------------------------------------------------------------------- one update
    def loss_and_grads(self, x, c, t=None, x0=None):
        """Forward + backward into the flat gradient buffer (all-reduced over the data-parallel group)."""
        self._check_alias()
        with _lib.on_device(self.flat_p):
            return self._loss_and_grads(x, c, t, x0)

    def _loss_and_grads(self, x, c, t, x0):
        lib = _lib.load()
        dev = self.flat_p.device
        x = _lib.require_cuda(x, "x")
        c = _lib.require_cuda(c, "c")
        B = x.shape[0]
        self._last_B = B
        need = self.net._get_plan().workspace_bytes(B, True)
        if self._ws is None or self._ws.numel() != need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ahead = self._ahead is not None and self._ahead == (self._ws.data_ptr(), B)
        if self._ahead is not None and not ahead:
            self.finish()  # (another batch size or workspace than the update in flight prepared for)
        self._ahead = None
        if self.prepare_ahead and not ahead:
            prepare_operands(self.net, self.p_views, self._ws, B)
        if t is None:  # reference: CPU generator for t, device generator for x_0 (models/base_model.py:209-212)
            t_host = self.model.time_distribution.sample([B] + [1] * (x.dim() - 1))
            t = self._stage_t(t_host, dev) if self.async_t else t_host.to(dev, torch.float32, non_blocking=True)
        if x0 is None:
            x0 = torch.randn_like(x)
        t = _lib.require_cuda(t, "t").reshape(-1)
        x0 = _lib.require_cuda(x0, "x0")
        s = _lib.stream_ptr(dev)
        xt, target = torch.empty_like(x), torch.empty_like(x)
        # the step's scalars - loss, squared gradient norm, gradient norm - in one fresh 16-byte tensor: zeroed by the trajectory kernel, accumulated into by
        # the loss / norm kernels, the norm's root written by the update kernel: no fill, clone or sqrt launches (each 5 us of serial stream time)
        self._scal = torch.empty(4, dtype=torch.float32, device=dev)
        self.loss, self.gnorm_sq, self.gnorm = self._scal[0], self._scal[1], self._scal[2]
        _lib.check(lib.v4h_cfm_prepare_z(_lib.ptr(x), _lib.ptr(x0), _lib.ptr(t), _lib.ptr(xt), _lib.ptr(target), B, x[0].numel(), s, _lib.ptr(self.loss),
                                         _lib.ptr(self.gnorm_sq)), "v4h_cfm_prepare_z")
        v, ws = run_forward(self.net, self.p_views, xt, t, c, True, ws=self._ws, reuse_operands=self.prepare_ahead or ahead)
        dv = torch.empty_like(v)
        _lib.check(lib.v4h_mse_loss_acc(_lib.ptr(v), _lib.ptr(target), _lib.ptr(self.loss), _lib.ptr(dv), v.numel(), s), "v4h_mse_loss_acc")
        W = world()
        if W > 1:  # DDP averages gradients: fold 1/world into the seed, then SUM
            _lib.check(lib.v4h_axpby(_lib.ptr(dv), _lib.ptr(dv), _lib.ptr(dv), 1.0 / W, 0.0, dv.numel(), s), "v4h_axpby")
        # The gradient buffer is this class's own and is used for nothing else: the backward WRITES every gradient (v4h_plan_set_gradient_mode 1: the
        # reduce pass of a weight gradient's split-K partials stores instead of adding, the tensors accumulated into are zeroed by the pass itself) -
        # no 104 MB zero fill per step, no read-modify-write of zeros.  The padding between tensors keeps its zeros from _flatten().  The plan is shared
        # with the autograd node (which accumulates into caller-owned tensors): the mode is set for this pass only.
        plan_h = self.net._get_plan().handle
        overwrite = self.overwrite_grads
        if overwrite:
            _lib.check(lib.v4h_plan_set_gradient_mode(plan_h, 1), "v4h_plan_set_gradient_mode")
        else:
            self.flat_g.zero_()
        try:
            self._backward(dv, ws)
        finally:
            if overwrite:
                lib.v4h_plan_set_gradient_mode(plan_h, 0)
        return self.loss

    def _backward(self, dv, ws):
        lib = _lib.load()
        if collectives_enabled() and os.environ.get("V4H_STAGED_CALLS") != "1":
            # one call; the library records an event when a stage's gradient slice is final and the bucket is reduced behind it
            if self.stage_events is None:
                self.stage_events = self.reducer.make_stage_events(len(self.stage_slices))
            # While buckets are in flight the persistent kernels can leave CUs to the collective's workgroups (read at enqueue time, so it brackets
            # exactly this pass; results do not depend on it).  Rehearsed on one GPU it is neutral (-0.9 ... +1.0 % over two runs of the table in
            # profiles/r03_comm_interference.md), so the default is 0; the switch is for runs beside the real RCCL kernels.
            _lib.check(lib.v4h_reserve_compute_units(self.comm_reserve_cus), "v4h_reserve_compute_units")
            try:
                run_backward_events(self.net, self.p_views, self.g_views, dv, ws, self.stage_events)
            finally:
                lib.v4h_reserve_compute_units(0)
            for (lo, hi), ev in zip(self.stage_slices, self.stage_events):
                self.reducer.reduce_slice_after(lo, hi, ev)
            self.reducer.finish()
        elif collectives_enabled():  # A/B hook: one call per stage, each ending with a join of the library's two streams
            for st, (lo, hi) in enumerate(self.stage_slices):
                run_backward(self.net, self.p_views, self.g_views, dv, ws, st, st)
                self.reducer.reduce_slice(lo, hi)
            self.reducer.finish()
        else:  # single rank: one call, so the weight-gradient stream is joined only once at the end
            run_backward(self.net, self.p_views, self.g_views, dv, ws, 0, len(self.stage_slices) - 1)

    def step(self, x, c, t=None, x0=None):
        """One BaseExperiment._step.  Returns (loss, grad_norm) as 0-dim device tensors (pre-clip norm, like clip_grad_norm_)."""
        with _lib.on_device(self.flat_p):
            if self.max_steps_ahead > 0 and not getattr(self, "_in_capture", False):
                self._throttle()
            if self.use_graph and t is None and x0 is None and self.max_grad_norm is None and not collectives_enabled():
                return self._step_graphed(x, c)
            return self._step(x, c, t, x0)

    # ---------------------------------------------------------------------------------------------- whole step as one hipGraph (opt-in)
    def _step_graphed(self, x, c):
        """The whole update - noise, trajectory, forward, two-stream backward, norm, AdamW - captured once into a hipGraph and replayed: the optimizer's
        step index and LR position live on the device (v4h_adamw_step_sched), so no kernel argument changes between replays; only t (sampled on the host
        generator like the reference, models/base_model.py:209-211) is copied into its static buffer first.  The returned tensors are the graph's own
        (overwritten by the next replay).  Opt-in (V4H_STEP_GRAPH=1 / use_graph): measured against the eager launch sequence in docs/history_r01-r04.md section 5."""
        self._check_alias()
        x = _lib.require_cuda(x, "x")
        c = _lib.require_cuda(c, "c")
        key = (tuple(x.shape), tuple(c.shape), self.flat_p.data_ptr())
        g = self._graph
        if g is None or g["key"] != key:
            if self._graph_warm < 2 or (g is not None and g["key"] != key):  # lazy initialisation (streams, LDS attributes, workspace) outside a capture
                self._graph_warm += 1
                self._graph = None
                return self._step(x, c, None, None)
            B = x.shape[0]
            g = {"key": key, "x": x.clone(), "c": c.clone(), "t": torch.empty([B] + [1] * (x.dim() - 1), dtype=torch.float32, device=x.device)}
            g["t"].copy_(self.model.time_distribution.sample(list(g["t"].shape)))
            self._in_capture = True
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    g["out"] = self._step(g["x"], g["c"], g["t"], None)
            finally:
                self._in_capture = False
            g["graph"] = graph
            self._graph = g
            # (the capture enqueued nothing: the counters the captured _step advanced on the host are advanced for real by the replay below)
            self.step_count -= 1
            self.iteration -= 1
        if g["x"].data_ptr() != x.data_ptr():
            g["x"].copy_(x, non_blocking=True)
        if g["c"].data_ptr() != c.data_ptr():
            g["c"].copy_(c, non_blocking=True)
        g["t"].copy_(self.model.time_distribution.sample(list(g["t"].shape)), non_blocking=True)
        g["graph"].replay()
        self.step_count += 1
        self.iteration += 1
        self.net.weights_epoch += 1
        if self.nonfinite_check_every > 0 and self.step_count % self.nonfinite_check_every == 0:
            self.raise_if_nonfinite()
        return g["out"]

    def _step(self, x, c, t, x0):
        lib = _lib.load()
        loss = self.loss_and_grads(x, c, t, x0)
        s = _lib.stream_ptr(self.flat_p.device)
        _lib.check(lib.v4h_sq_norm_accum(_lib.ptr(self.flat_g), self.total, _lib.ptr(self.gnorm_sq), s), "v4h_sq_norm_accum")
        self.step_count += 1
        # `step > MIN_STEP_SKIP` of the reference is the 0-based index of the training loop (base_experiment.py:475-479,586)
        skip_above = self.max_grad_norm if (self.max_grad_norm is not None and self.iteration > self.MIN_STEP_SKIP) else float("inf")
        self.iteration += 1
        st_in, st_out = self._state[self._cur], self._state[self._cur ^ 1]
        capturing = getattr(self, "_in_capture", False)
        if self.pipeline_update and not capturing and not self.use_graph and not collectives_enabled() and self._ws is not None and self._last_B is not None:
            hyper = (self.clip if self.clip is not None else float("inf"), self.lr, self.eta_min, self.iterations, self.betas[0], self.betas[1], self.eps, self.wd,
                     skip_above)
            update_ahead(self.net, self.p_views, (self.flat_p, self.flat_g, self.flat_m, self.flat_v), self._offsets_c, self._ws, self._last_B, self.gnorm_sq,
                         hyper, st_in, st_out, self.nonfinite, self.gnorm)
            self._ahead = (self._ws.data_ptr(), self._last_B)
            self._cur ^= 1
            self.net.weights_epoch += 1
            out_loss = loss
            if self.nonfinite_check_every > 0 and self.step_count % self.nonfinite_check_every == 0:
                self.raise_if_nonfinite()
            # The norm's root is written by the update's first launch on the library's SIDE stream, which the current stream does not wait for: hand back the
            # squared norm's root computed on the current stream instead (one 1-element launch, same value: sqrt of the very scalar the update reads).
            return out_loss, torch.sqrt(self.gnorm_sq)
        args = (_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v), self.total, _lib.ptr(self.gnorm_sq),
                self.clip if self.clip is not None else float("inf"), self.lr, self.eta_min, self.iterations, self.betas[0], self.betas[1], self.eps, self.wd,
                _lib.ptr(st_in), _lib.ptr(st_out), skip_above, s, _lib.ptr(self.nonfinite), _lib.ptr(self.gnorm))
        if self.flat_ema is not None:
            _lib.check(lib.v4h_adamw_step_sched_ema(*args, _lib.ptr(self.flat_ema), self.ema_decay), "v4h_adamw_step_sched_ema")
        else:
            _lib.check(lib.v4h_adamw_step_sched(*args), "v4h_adamw_step_sched")
        if capturing:  # a replayed graph has fixed pointers: copy the new counters back instead of swapping the two buffers
            st_in.copy_(st_out)
        else:
            self._cur ^= 1
        self.net.weights_epoch += 1  # parameters rewritten through raw pointers: invalidate cached operand copies (ViT.operands_current)
        out_loss = loss  # (this step's own scalar tensor: nothing else writes it)
        if collectives_enabled():
            dist.all_reduce(out_loss, op=dist.ReduceOp.SUM, group=self.group)
            out_loss /= world()
        if not capturing and self.nonfinite_check_every > 0 and self.step_count % self.nonfinite_check_every == 0:
            self.raise_if_nonfinite()
        return out_loss, self.gnorm

    def _throttle(self):
        """Keep the host at most ``max_steps_ahead`` update steps ahead of the device (see __init__)."""
        ring = self._ahead_events
        ev = None
        if len(ring) >= self.max_steps_ahead:
            ev = ring.pop(0)
            ev.synchronize()
        if ev is None:
            ev = torch.cuda.Event()  # (the ring's events are re-used: no event is created after the first max_steps_ahead steps)
        ev.record()
        ring.append(ev)

    def finish(self):
        """Order a pipelined update (pipeline_update) into the current stream: call before parameters, moments or gradients are read by anything but the
        next step().  A no-op otherwise."""
        if self._ahead is not None:
            plan_join(self.net, self.flat_p.device)
            self._ahead = None

    def raise_if_nonfinite(self):
        """Host look at the sticky device counter of skipped (non-finite) updates: raise like the reference.  The optimizer's step index and the LR
        schedule live on the device and never advanced for a skipped update; the host's ``step_count`` is re-read from there."""
        self.finish()
        skipped = int(self.nonfinite.item())
        self.sync_counters()
        if skipped:
            self.nonfinite.zero_()
            raise RuntimeError(f"{self.NONFINITE_MSG} ({skipped} update(s) skipped on the device; step_count is {self.step_count})")

    def sync_counters(self):
        """One 16-byte read: {applied optimizer steps, scheduler steps, updates skipped for max_grad_norm}; brings ``step_count`` in line."""
        self.finish()
        a, k, skipped, _ = (int(v) for v in self._state[self._cur].tolist())
        self.step_count = a
        return {"optimizer_steps": a, "scheduler_steps": k, "skipped_max_grad_norm": skipped}

    @property
    def current_lr(self):
        """Learning rate of the NEXT update (``optimizer.param_groups[0]["lr"]`` of the reference); syncs."""
        return self.eta_min + (self.lr - self.eta_min) * 0.5 * (1.0 + math.cos(math.pi * self.sync_counters()["scheduler_steps"] / self.iterations))

    # ---------------------------------------------------------------------------------------------- checkpoints (base_experiment.py:661-677)
    def _torch_param_order(self):
        """Indices into the flat layout (the C ABI's parameter order) in ``model.parameters()`` order - what torch.optim.AdamW(model.parameters())
        numbers 0..n-1.  Both orders are the state_dict() order for the shipped networks; computed rather than assumed."""
        by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        order = [by_ptr[q.data_ptr()] for q in self.model.parameters() if q.data_ptr() in by_ptr]
        if sorted(order) != list(range(len(self.params))):
            raise RuntimeError("CFMTrainer: model.parameters() and the network's parameter list differ (a parameter outside the ViT?)")
        return order

    def optimizer_state_dict(self):
        """``torch.optim.AdamW.state_dict()`` of this run: loadable by the reference's optimizer (base_experiment.py:374-388)."""
        self._check_alias()
        cnt = self.sync_counters()
        state = {}
        for k, i in enumerate(self._torch_param_order()):
            o, n = self.offsets[i], self.params[i].numel()
            state[k] = {"step": torch.tensor(float(cnt["optimizer_steps"])),
                        "exp_avg": self.flat_m[o : o + n].view_as(self.params[i]).clone(),
                        "exp_avg_sq": self.flat_v[o : o + n].view_as(self.params[i]).clone()}
        group = {"lr": self.current_lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True, "initial_lr": self.lr,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def scheduler_state_dict(self):
        """``CosineAnnealingLR.state_dict()`` at this run's position (base_experiment.py:403-408)."""
        k = self.sync_counters()["scheduler_steps"]
        lr = self.current_lr
        return {"T_max": self.iterations, "eta_min": self.eta_min, "base_lrs": [self.lr], "last_epoch": k, "_step_count": k + 1, "_is_initial": False,
                "_get_lr_called_within_step": False, "_last_lr": [lr]}

    def state_dict(self):
        """The trainer's half of the reference's checkpoint file: ``{"optimizer": ..., "scheduler": ...}`` (add ``"model": model.state_dict()``)."""
        return {"optimizer": self.optimizer_state_dict(), "scheduler": self.scheduler_state_dict()}

    def ema_state_dict(self):
        """``torch_ema.ExponentialMovingAverage.state_dict()`` of this run (base_experiment.py:674): decay, num_updates (= applied optimizer steps: the
        reference calls ema.update() exactly once per applied update), shadow_params in ``model.parameters()`` order, collected_params."""
        if self.flat_ema is None:
            return None
        self._check_alias()
        n = self.sync_counters()["optimizer_steps"] - getattr(self, "_ema_step0", 0)
        shadow = []
        for i in self._torch_param_order():
            o, k = self.offsets[i], self.params[i].numel()
            shadow.append(self.flat_ema[o : o + k].view_as(self.params[i]).clone())
        return {"decay": self.ema_decay, "num_updates": n, "shadow_params": shadow, "collected_params": None}

    def load_ema_state_dict(self, sd):
        if self.flat_ema is None:
            raise RuntimeError("CFMTrainer: constructed without ema_decay")
        self.finish()
        self._check_alias()
        order = self._torch_param_order()
        if len(sd["shadow_params"]) != len(order):
            raise ValueError("shadow_params must have the same length as the parameters")  # torch_ema's check
        self.ema_decay = float(sd["decay"])
        for k, i in enumerate(order):
            o, n = self.offsets[i], self.params[i].numel()
            self.flat_ema[o : o + n].view_as(self.params[i]).copy_(sd["shadow_params"][k])
        # the kernel derives torch_ema's warm-up count from the optimizer's step counter: remember the difference (0 for a file written by this class or
        # by the reference, where both count the same updates)
        self._ema_step0 = self.sync_counters()["optimizer_steps"] - int(sd["num_updates"] or 0)
        if self._ema_step0 != 0:
            raise NotImplementedError("CFMTrainer.load_ema_state_dict: num_updates differs from the optimizer's step count (the fused kernel takes the EMA "
                                      "warm-up position from the optimizer step); load the optimizer state first, or use the autograd route")

    class _Averaged:
        def __init__(self, tr):
            self.tr = tr

        def __enter__(self):  # torch_ema: store(), copy_to()
            tr = self.tr
            tr.finish()
            tr._check_alias()
            tr._ema_collected = tr.flat_p.clone()
            tr.flat_p.copy_(tr.flat_ema)
            tr.net.weights_epoch += 1
            return tr

        def __exit__(self, *exc):  # torch_ema: restore()
            tr = self.tr
            tr.flat_p.copy_(tr._ema_collected)
            tr._ema_collected = None
            tr.net.weights_epoch += 1
            return False

    def average_parameters(self):
        """``with trainer.average_parameters(): validate()`` - the model runs with the shadow parameters inside the block (base_experiment.py:630-631)."""
        if self.flat_ema is None:
            raise RuntimeError("CFMTrainer: constructed without ema_decay")
        return CFMTrainer._Averaged(self)

    def checkpoint(self):
        """The whole file the reference's ``_save_model`` writes (base_experiment.py:667-675)."""
        return {"model": self.model.state_dict(), **self.state_dict(), "ema": self.ema_state_dict()}

    def load_state_dict(self, sd):
        """Continue from ``state_dict()`` / ``checkpoint()`` of this class, or from the "optimizer" / "scheduler" entries written by the reference's
        torch.optim.AdamW + CosineAnnealingLR (a "model" entry, when present, is loaded into the model first)."""
        self.finish()
        self._check_alias()
        if "model" in sd and sd["model"] is not None:
            self.model.load_state_dict(sd["model"])  # in place: the parameters stay views of the flat buffer
            self.net.weights_epoch += 1
        opt, sch = sd.get("optimizer"), sd.get("scheduler")
        applied = sched = None
        if opt is not None:
            groups = opt["param_groups"]
            if len(groups) != 1:
                raise NotImplementedError("CFMTrainer.load_state_dict: one parameter group only (per-module learning rates train through the autograd node)")
            g = groups[0]
            if g.get("amsgrad") or g.get("maximize"):
                raise NotImplementedError("CFMTrainer.load_state_dict: amsgrad / maximize optimizers are not this update rule")
            order = self._torch_param_order()
            if len(g["params"]) != len(order):
                raise ValueError(f"optimizer state has {len(g['params'])} parameters, the network {len(order)}")
            self.betas, self.eps, self.wd = (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"]), float(g["weight_decay"])
            self.lr = float(g.get("initial_lr", g["lr"]))
            steps = set()
            self.flat_m.zero_()
            self.flat_v.zero_()
            for k, i in enumerate(order):
                stt = opt["state"].get(g["params"][k])
                if stt is None:  # never updated: fresh moments
                    continue
                o, n = self.offsets[i], self.params[i].numel()
                if tuple(stt["exp_avg"].shape) != tuple(self.params[i].shape):
                    raise ValueError(f"optimizer state {k}: shape {tuple(stt['exp_avg'].shape)} does not match parameter {tuple(self.params[i].shape)}")
                self.flat_m[o : o + n].view_as(self.params[i]).copy_(stt["exp_avg"])
                self.flat_v[o : o + n].view_as(self.params[i]).copy_(stt["exp_avg_sq"])
                steps.add(int(float(stt["step"])))
            if len(steps) > 1:
                raise NotImplementedError(f"CFMTrainer.load_state_dict: parameters with different step counts {sorted(steps)} (one fused update for all)")
            applied = steps.pop() if steps else 0
        if sch is not None:
            self.iterations, self.eta_min = int(sch["T_max"]), float(sch["eta_min"])
            self.lr = float(sch["base_lrs"][0])
            sched = int(sch["last_epoch"])
        cur = self._state[self._cur].tolist()
        new = [cur[0] if applied is None else applied, cur[1] if sched is None else sched, cur[2], 0]
        if applied is not None and sched is None:
            new[1] = applied
        for t in self._state:
            t.copy_(torch.tensor(new, dtype=torch.int32))
        if sd.get("ema") is not None and self.flat_ema is not None:
            self.load_ema_state_dict(sd["ema"])
        self.nonfinite.zero_()
        self.step_count = new[0]
        self._graph = None  # a captured step (use_graph) has the old hyper-parameters as kernel arguments: capture again

    @staticmethod
    def check_finite(grad_norm):
        """error_if_nonfinite=True of the reference's clip_grad_norm_ call (base_experiment.py:581); needs a host sync."""
        g = float(grad_norm)
        if not math.isfinite(g):
            raise RuntimeError(CFMTrainer.NONFINITE_MSG)
        return g
