// Host-side runtime of libvit4hep_hip.so: plan (derived sizes + HBM workspace layout), the forward and the staged
// backward launch sequences of CaloChallengeCFM.forward (calochallenge_cfm/model.py:62-66 -> nn/vit.py:185-206), and the
// C ABI of include/vit4hep_hip.h.  No device allocation, no synchronisation: everything is enqueued on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/vit4hep_hip.h"
#include "v4h_ops.h"

using namespace v4h;

// ------------------------------------------------------------------------------------------------ error slot
static thread_local char g_err[512] = "";
void v4h_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------ plan
enum {  // parameter indices in state_dict() order (nn/vit.py:76-132)
  P_FREQS = 0, P_XW, P_XB, P_C0W, P_C0B, P_C2W, P_C2B, P_T0W, P_T0B, P_T2W, P_T2B, P_BLOCK0,
  B_QKVW = 0, B_QKVB, B_PROJW, B_PROJB, B_FC1W, B_FC1B, B_FC2W, B_FC2B, B_ADAW, B_ADAB, B_COUNT,
  F_LINW = 0, F_LINB, F_ADAW, F_ADAB, F_COUNT
};

struct v4h_plan {
  v4h_config cfg;
  Mode mode;
  int T, P, Ppad, D, H, DH, M, Kc, Kcpad, F, depth;  // Kc: input width of c_embedder.0 (= condition_dim without a condition mapper)
  int Kcx, Kcxpad;  // width of the conditions the caller passes: Kc, or condition_dim in front of a condition-embedding mapper
  int Px, Pxpad;  // input width of the x_embedder Linear: P, or x_embed_in behind an embedding mapper
  bool mapper() const { return cfg.x_embed_in > 0; }
  int ldmod() const { return 6 * D * depth + 2 * D; }  // row stride of the modulation table: [block 0: 6 D | ... | block depth-1: 6 D | final layer: 2 D]
  bool cmapper() const { return cfg.c_embed_in > 0; }
  int cmw() const { return nparams() - 2; }  // condition mapper weight / bias: the last two tensors
  int cmb() const { return nparams() - 1; }
  int xmw() const { return nparams() - 2 - (cmapper() ? 2 : 0); }  // x mapper weight / bias: before them
  int xmb() const { return nparams() - 1 - (cmapper() ? 2 : 0); }
  PatchGeom pg;
  bool mapped = false;  // general geometry: gather / scatter through a caller-provided index map, positions from a caller-provided table
  long V = 0;           // voxels per sample
  std::vector<int> rows, cols;  // per parameter; cols == 0 for 1-D tensors
  int nparams() const { return (int)rows.size(); }
  int blk(int i, int k) const { return P_BLOCK0 + B_COUNT * i + k; }
  int fin(int k) const { return P_BLOCK0 + B_COUNT * depth + k; }
  // Side stream for the weight-gradient contractions of the backward pass (created on first use): wgrad and dgrad of a
  // Linear both consume dY and are independent, so they run concurrently and fill each other's idle CUs / tile tails.
  // Fork/join with events only, so the caller's stream ordering (and graph capture) stays intact.
  mutable hipStream_t side = nullptr;
  mutable hipEvent_t ev[8] = {};
  mutable hipEvent_t evS[4] = {};  // side-stream progress marks (after the fc2 / fc1 / proj / qkv weight gradient of a block, after an adaLN backward)
  mutable hipEvent_t evOps = nullptr;   // operand copies made ahead of the next forward on the side stream (v4h_vit_prepare_operands)
  mutable bool ops_pending = false;     // ... and not yet waited for by a forward
  mutable unsigned mark_live = 0;       // side-stream marks recorded by the backward call in progress
  mutable std::vector<hipEvent_t> evUp; // pipelined update (v4h_vit_update_ahead): one event per stage, in the order the forward consumes the weights
  mutable unsigned long long upd_mask = 0;  // stages whose event the next forward still has to wait for
  mutable int evi = 0;
  mutable bool side_ok = false;
  mutable bool grad_overwrite = false;  // v4h_plan_set_gradient_mode: backward passes WRITE every gradient of their stages (caller need not zero)
  mutable bool x16 = false, g16 = false;  // v4h_plan_set_residual_storage: the residual stream / its gradient are kept in the mode type (bf16 mode)
  mutable int device = -1;  // device the side stream and events were created on (first forward / backward call)
};

static bool g_overlap_wgrad = true;  // V4H_WGRAD_OVERLAP=0 disables the side stream
static bool g_batch_adaln = true;    // V4H_BATCH_ADALN=0: per-block adaLN backward also in single-call passes (A/B hook)
static int side_init(const v4h_plan& p) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { v4h_set_error("hipGetDevice failed"); return V4H_ERR_HIP; }
  if (p.side_ok) {
    // the plan's side stream and events live on one device: a call made while another device is current would launch the weight gradients
    // there (invalid-handle errors or cross-device accesses).  Bind a plan to one device; the Python host sets it around every call.
    V4H_CHECK_ARG(dev == p.device, "plan is bound to device %d (its side stream and events) but device %d is current: set the device before the call, or use one plan per device", p.device, dev);
    return V4H_OK;
  }
  p.device = dev;
  const char* e = getenv("V4H_WGRAD_OVERLAP");
  if (e && e[0] == '0') g_overlap_wgrad = false;
  e = getenv("V4H_BATCH_ADALN");
  if (e && e[0] == '0') g_batch_adaln = false;
  // HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams that land on the same queue run
  // their kernels back to back.  Measured: after torch.distributed + a communication stream exist, a plain side stream shares the
  // main stream's queue and ALL overlap is lost (144.8 vs 166.9 steps/s).  Streams of another priority class get their own queue.
  const char* pe = getenv("V4H_SIDE_PRIORITY");
  int least = 0, greatest = 0;
  hipError_t se;
  if ((!pe || pe[0] != '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
    se = hipStreamCreateWithPriority(&p.side, hipStreamNonBlocking, (pe && pe[0] == 'l') ? least : greatest);  // V4H_SIDE_PRIORITY=low: the other non-default class
  else
    se = hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking);
  if (se != hipSuccess) { v4h_set_error("cannot create side stream"); return V4H_ERR_HIP; }
  // The events only order the two streams of this device against each other; nobody inspects them from the host, so recording one
  // needs no system-scope release (an L2 write-back + ~6 us bubble in front of the next kernel of the recording stream, 30 times per
  // step: 180.1 -> 182.8 steps/s).  The kernels' own agent-scope release at their end is what the other stream's kernels need.
  const unsigned evflags = hipEventDisableTiming | hipEventDisableSystemFence;
  for (int i = 0; i < 8; ++i)
    if (hipEventCreateWithFlags(&p.ev[i], evflags) != hipSuccess) { v4h_set_error("cannot create event"); return V4H_ERR_HIP; }
  for (int i = 0; i < 4; ++i)
    if (hipEventCreateWithFlags(&p.evS[i], evflags) != hipSuccess) { v4h_set_error("cannot create event"); return V4H_ERR_HIP; }
  if (hipEventCreateWithFlags(&p.evOps, evflags) != hipSuccess) { v4h_set_error("cannot create event"); return V4H_ERR_HIP; }
  p.side_ok = true;
  return V4H_OK;
}
// the side stream waits for everything enqueued on the main stream so far
static int side_wait_main(const v4h_plan& p, hipStream_t main) {
  hipEvent_t e = p.ev[p.evi++ & 7];
  if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(p.side, e, 0) != hipSuccess) { v4h_set_error("fork failed"); return V4H_ERR_HIP; }
  return V4H_OK;
}
// The side stream waits for the operator the caller enqueues on the main stream between arm_fork() and complete_fork() - and for nothing behind it: the
// operator's last launch carries the event as its completion signal (V4H_LAUNCH); an operator that did not take it gets the record of side_wait_main().
thread_local hipEvent_t v4h_tls_stop_event = nullptr;
static const bool g_stop_events = !(getenv("V4H_STOP_EVENTS") && getenv("V4H_STOP_EVENTS")[0] == '0');  // A/B hook
struct ForkArm { hipEvent_t e; bool armed; };
static ForkArm arm_fork(const v4h_plan& p, hipStream_t main) {
  ForkArm f{p.ev[p.evi++ & 7], false};
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(main, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;  // (a captured launch has no completion signal to hand out)
  f.armed = g_stop_events && !capturing;
  v4h_tls_stop_event = f.armed ? f.e : nullptr;
  return f;
}
static int complete_fork(const v4h_plan& p, const ForkArm& f, hipStream_t main, bool call_ok) {
  const bool taken = f.armed && v4h_tls_stop_event == nullptr;
  v4h_tls_stop_event = nullptr;
  if (!call_ok) return V4H_OK;  // (the operator failed: its error stands)
  if (!taken && hipEventRecord(f.e, main) != hipSuccess) { v4h_set_error("fork failed"); return V4H_ERR_HIP; }
  if (hipStreamWaitEvent(p.side, f.e, 0) != hipSuccess) { v4h_set_error("fork failed"); return V4H_ERR_HIP; }
  return V4H_OK;
}
// `waiter` waits for everything enqueued on `signaler` so far
static int stream_wait(const v4h_plan& p, hipStream_t waiter, hipStream_t signaler) {
  if (waiter == signaler) return V4H_OK;
  hipEvent_t e = p.ev[p.evi++ & 7];
  if (hipEventRecord(e, signaler) != hipSuccess || hipStreamWaitEvent(waiter, e, 0) != hipSuccess) { v4h_set_error("stream fork / join failed"); return V4H_ERR_HIP; }
  return V4H_OK;
}
// the main stream waits for everything enqueued on the side stream so far
static int main_wait_side(const v4h_plan& p, hipStream_t main) {
  hipEvent_t e = p.ev[p.evi++ & 7];
  if (hipEventRecord(e, p.side) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) { v4h_set_error("join failed"); return V4H_ERR_HIP; }
  return V4H_OK;
}

// Lagged joins.  The weight-gradient stream only READS the backward temporaries (dy, dy2, dhpre, dqkv); the main stream must not
// overwrite one before its reader has finished.  Joining the streams at every such point stalls the main stream for the side
// stream's tail plus the cross-stream signal latency (7-13 us each, 12 per step).  Instead the temporaries exist twice (consecutive
// blocks alternate), the side stream drops a mark behind a block's last weight gradient, and the main stream waits for that mark
// two blocks later, when it has long been reached: one wait packet per block in the main queue.
enum { S_FC2 = 0, S_ADA, S_BLK0, S_BLK1 };
static int side_mark(const v4h_plan& p, int which) {
  if (hipEventRecord(p.evS[which], p.side) != hipSuccess) { v4h_set_error("mark failed"); return V4H_ERR_HIP; }
  p.mark_live |= 1u << which;
  return V4H_OK;
}
// Only marks dropped by THIS call are waited for: every backward call ends with a full join of the two streams, so an older mark is long reached - and a
// stream that is being captured into a hipGraph must not wait for an event recorded outside the capture.
static int main_wait_mark(const v4h_plan& p, int which, hipStream_t main) {
  if (!(p.mark_live & (1u << which))) return V4H_OK;
  if (hipStreamWaitEvent(main, p.evS[which], 0) != hipSuccess) { v4h_set_error("wait for mark failed"); return V4H_ERR_HIP; }
  return V4H_OK;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int32_t v4h_abi_version(void) { return V4H_ABI_VERSION; }
extern "C" const char* v4h_last_error(void) { return g_err; }

static int plan_create_impl(const v4h_config* c, bool mapped, int tokens, int patch_dim, long voxels, v4h_plan** out) {
  V4H_CHECK_ARG(c && out, "plan_create: null argument");
  V4H_CHECK_ARG(c->mode == V4H_MODE_F32 || c->mode == V4H_MODE_BF16, "plan_create: unknown mode %d", c->mode);
  V4H_CHECK_ARG(c->in_channels == 1, "plan_create: in_channels %d unsupported (every shape-CFM config uses 1)", c->in_channels);
  if (!mapped) {
    for (int k = 0; k < 3; ++k)
      V4H_CHECK_ARG(c->shape[k] > 0 && c->patch_shape[k] > 0 && c->shape[k] % c->patch_shape[k] == 0,
                    "Input size (%d) should be divisible by patch size (%d) in axis %d.", c->shape[k], c->patch_shape[k], k);
  } else {
    V4H_CHECK_ARG(tokens > 0 && patch_dim > 0 && voxels > 0, "plan_create_mapped: tokens %d / patch_dim %d / voxels %ld must be positive", tokens, patch_dim, voxels);
    V4H_CHECK_ARG(voxels < (1L << 31), "plan_create_mapped: %ld voxels per sample do not fit the int32 index map", voxels);
  }
  V4H_CHECK_ARG(c->depth >= 1, "plan_create: depth %d", c->depth);
  V4H_CHECK_ARG(c->hidden_dim % c->num_heads == 0, "dim should be divisible by num_heads");
  V4H_CHECK_ARG(c->hidden_dim % 96 == 0 || c->hidden_dim % 32 == 0, "plan_create: hidden_dim %d must be a multiple of 32", c->hidden_dim);
  V4H_CHECK_ARG(c->hidden_dim % 6 == 0, "plan_create: hidden_dim %d must be divisible by 6 (3-D sincos embedding)", c->hidden_dim);
  V4H_CHECK_ARG(c->hidden_dim / c->num_heads == 80, "plan_create: head_dim %d not built (only 80)", c->hidden_dim / c->num_heads);
  V4H_CHECK_ARG(c->mlp_hidden % 32 == 0 && c->freq_dim % 32 == 0, "plan_create: mlp_hidden/freq_dim must be multiples of 32");
  v4h_plan* p = new v4h_plan();
  p->cfg = *c;
  p->mode = (Mode)c->mode;
  p->mapped = mapped;
  if (!mapped) {
    p->pg = PatchGeom{c->shape[0], c->shape[1], c->shape[2], c->patch_shape[0], c->patch_shape[1], c->patch_shape[2],
                      c->shape[0] / c->patch_shape[0], c->shape[1] / c->patch_shape[1], c->shape[2] / c->patch_shape[2]};
    p->T = p->pg.l * p->pg.a * p->pg.r;
    p->P = c->patch_shape[0] * c->patch_shape[1] * c->patch_shape[2];
    p->V = (long)c->shape[0] * c->shape[1] * c->shape[2];
  } else {
    p->pg = PatchGeom{};
    p->T = tokens;
    p->P = patch_dim;
    p->V = voxels;
  }
  p->Ppad = round_up(p->P, 32);
  V4H_CHECK_ARG(c->x_embed_in >= 0, "plan_create: x_embed_in %d", c->x_embed_in);
  p->Px = c->x_embed_in > 0 ? c->x_embed_in : p->P;
  p->Pxpad = round_up(p->Px, 32);
  p->D = c->hidden_dim;
  p->H = c->num_heads;
  p->DH = p->D / p->H;
  p->M = c->mlp_hidden;
  V4H_CHECK_ARG(c->c_embed_in >= 0, "plan_create: c_embed_in %d", c->c_embed_in);
  p->Kcx = c->condition_dim;
  p->Kcxpad = round_up(p->Kcx, 32);
  p->Kc = c->c_embed_in > 0 ? c->c_embed_in : c->condition_dim;
  p->Kcpad = round_up(p->Kc, 32);
  p->F = c->freq_dim;
  p->depth = c->depth;
  auto add = [&](int r, int cc) { p->rows.push_back(r); p->cols.push_back(cc); };
  const int D = p->D;
  add(D / 6, 0);
  add(D, p->Px); add(D, 0);
  add(D, p->Kc); add(D, 0); add(D, D); add(D, 0);
  add(D, p->F); add(D, 0); add(D, D); add(D, 0);
  for (int i = 0; i < p->depth; ++i) {
    add(3 * D, D); add(3 * D, 0); add(D, D); add(D, 0);
    add(p->M, D); add(p->M, 0); add(D, p->M); add(D, 0);
    add(6 * D, D); add(6 * D, 0);
  }
  add(p->P, D); add(p->P, 0); add(2 * D, D); add(2 * D, 0);
  if (p->mapper()) { add(p->Px, p->P); add(p->Px, 0); }
  if (p->cmapper()) { add(p->Kc, p->Kcx); add(p->Kc, 0); }
  *out = p;
  return V4H_OK;
}
extern "C" int32_t v4h_plan_create(const v4h_config* c, v4h_plan** out) { return plan_create_impl(c, false, 0, 0, 0, out); }
extern "C" int32_t v4h_plan_create_mapped(const v4h_config* c, int32_t tokens, int32_t patch_dim, int64_t voxels, v4h_plan** out) {
  return plan_create_impl(c, true, tokens, patch_dim, (long)voxels, out);
}
// geometry-dependent steps: regular grid (closed-form index arithmetic) or caller-provided tables
static int check_geom(const v4h_plan* p, const int32_t* map, const float* pos, const char* who) {
  if (p->mapped) V4H_CHECK_ARG(map != nullptr && pos != nullptr, "%s: a plan made by v4h_plan_create_mapped needs d_patch_map and d_pos", who);
  else V4H_CHECK_ARG(map == nullptr, "%s: d_patch_map given to a regular-grid plan (use v4h_plan_create_mapped)", who);
  return V4H_OK;
}
extern "C" void v4h_plan_destroy(v4h_plan* p) {
  if (p && p->side_ok) {
    for (int i = 0; i < 8; ++i) hipEventDestroy(p->ev[i]);
    for (int i = 0; i < 4; ++i) hipEventDestroy(p->evS[i]);
    hipEventDestroy(p->evOps);
    for (hipEvent_t e : p->evUp) hipEventDestroy(e);
    hipStreamDestroy(p->side);
  }
  delete p;
}
extern "C" int32_t v4h_plan_num_params(const v4h_plan* p) { return p ? p->nparams() : 0; }
extern "C" int32_t v4h_plan_param_shape(const v4h_plan* p, int32_t i, int32_t* r, int32_t* c) {
  V4H_CHECK_ARG(p && i >= 0 && i < p->nparams(), "param_shape: bad index %d", i);
  *r = p->rows[i];
  *c = p->cols[i];
  return V4H_OK;
}
extern "C" int32_t v4h_vit_num_backward_stages(const v4h_plan* p) { return p ? p->depth + 2 : 0; }

// ------------------------------------------------------------------------------------------------ workspace layout
struct BlockWS {
  float *mean1, *rstd1, *mean2, *rstd2, *lse;
  char *x_mid;  // residual stream between the two branches: f32, or the mode type (plan.x16)
  char *u1, *qkv, *o, *y1, *u2, *hgrad, *h, *y2;
};
struct WS {
  std::vector<char*> wop;       // operand-typed (cast / padded) weights, null where the f32 parameter itself is used
  float* linb_pad;
  char *xp, *temb, *ht, *cpad, *hc, *silu_c, *uf;
  char *xpm, *xmb_pad, *dxpre;   // embedding mapper (fine-tuning): gathered input patches, padded bias, gradient of the pre-activation
  float *xpre, *gxmw, *gxmb;
  float *pe, *ht_pre, *hc_pre, *cond, *cemb, *modf, *meanf, *rstdf;
  float *mod_all, *adaB;   // every adaLN modulation of the step in one table (B x ldmod); concatenated adaLN biases
  char* adaW;              // concatenated operand copies of the adaLN weights (ldmod x D), bf16 mode: one contraction makes the whole table
  std::vector<float*> mod;
  std::vector<char*> X;  // residual stream in front of block i (X[depth]: in front of the final layer): f32, or the mode type (plan.x16)
  std::vector<BlockWS> blk;
  // backward
  char* zero_begin; size_t zero_bytes;
  char *cin, *cmb_pad, *dcpre;  // condition mapper: its input operand, padded bias, d pre-activation
  float *cpre, *gcmw, *gcmb;
  float* dmod_base;
  float** gtab;  // device table of the grouped adaLN weight-gradient contraction
  float *dsilu, *gxw, *gc0w, *glin, *glinb;
  char *dxA, *dxB;  // residual-stream gradient ping-pong: f32, or the mode type (plan.g16)
  float *delta, *G, *slab[2];  // slab[0]: main stream, slab[1]: side stream
  char *dy[2], *dy2[2], *dhpre[2], *dqkv[2];
  char *dvp, *du, *dof, *dmod_t, *dcond, *dh_small, *dh_small2, *dx0_t;  // dy: gradient entering the MLP half of a block, dy2: the attention half
  size_t total;
};

// scratch for the split-K partials of the largest weight gradient: at most 16 splits (gemm_wgrad_splitk) of the fc1 / fc2 or qkv shape
static size_t slab_bytes(const v4h_plan& p) {
  const size_t D = p.D, M = p.M;
  size_t mx = 16 * D * M;
  if (16 * 3 * D * D > mx) mx = 16 * 3 * D * D;
  return mx * 4;
}

static void layout(const v4h_plan& p, int B, bool training, char* base, WS& w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return r;
  };
  const size_t es = esize(p.mode);
  const size_t BT = (size_t)B * p.T, D = p.D, M = p.M;
  w.wop.assign(p.nparams(), nullptr);
  w.adaW = p.mode == MODE_BF16 ? take((size_t)p.ldmod() * D * es) : nullptr;
  w.adaB = (float*)take((size_t)p.ldmod() * 4);
  if (w.adaW || !base) {
    for (int i = 0; i < p.depth && p.mode == MODE_BF16; ++i) w.wop[p.blk(i, B_ADAW)] = base ? w.adaW + (size_t)i * 6 * D * D * es : (char*)1;
    if (p.mode == MODE_BF16) w.wop[p.fin(F_ADAW)] = base ? w.adaW + (size_t)p.depth * 6 * D * D * es : (char*)1;
  }
  for (int i = 0; i < p.nparams(); ++i) {
    if (p.cols[i] == 0 || w.wop[i]) continue;
    int rp = p.rows[i], cp = p.cols[i];
    bool padded = false;
    if (i == P_XW) { cp = p.Pxpad; padded = true; }
    if (p.mapper() && i == p.xmw()) { rp = p.Pxpad; cp = p.Ppad; padded = true; }
    if (i == P_C0W) { cp = p.Kcpad; padded = true; }
    if (p.cmapper() && i == p.cmw()) { rp = p.Kcpad; cp = p.Kcxpad; padded = true; }
    if (i == p.fin(F_LINW)) { rp = p.Ppad; padded = true; }
    if (p.mode == MODE_BF16 || padded) w.wop[i] = take((size_t)rp * cp * es);
  }
  w.linb_pad = (float*)take(p.Ppad * 4);
  w.xp = take(BT * p.Pxpad * es);
  w.xpm = w.xmb_pad = nullptr; w.xpre = nullptr;
  if (p.mapper()) {
    w.xpm = take(BT * p.Ppad * es);                    // gathered patches of the new geometry (mapper input)
    w.xpre = (float*)take(BT * p.Pxpad * 4);           // mapper pre-activation (for silu')
    w.xmb_pad = take((size_t)p.Pxpad * 4);             // mapper bias, zero-padded
  }
  w.pe = (float*)take((size_t)p.T * D * 4);
  w.temb = take((size_t)B * p.F * es);
  w.ht_pre = (float*)take((size_t)B * D * 4);
  w.ht = take((size_t)B * D * es);
  w.cpad = take((size_t)B * p.Kcpad * es);
  w.cin = nullptr; w.cpre = nullptr; w.cmb_pad = nullptr;
  if (p.cmapper()) {
    w.cin = take((size_t)B * p.Kcxpad * es);           // the caller's conditions (mapper input)
    w.cpre = (float*)take((size_t)B * p.Kcpad * 4);    // mapper pre-activation (for silu')
    w.cmb_pad = take((size_t)p.Kcpad * 4);             // mapper bias, zero-padded
  }
  w.hc_pre = (float*)take((size_t)B * D * 4);
  w.hc = take((size_t)B * D * es);
  w.cond = (float*)take((size_t)B * D * 4);
  w.cemb = (float*)take((size_t)B * D * 4);  // c_embedder output (kept across evaluations with the same conditions)
  w.silu_c = take((size_t)B * D * es);
  w.mod.resize(p.depth);
  w.mod_all = (float*)take((size_t)B * p.ldmod() * 4);
  for (int i = 0; i < p.depth; ++i) w.mod[i] = w.mod_all ? w.mod_all + (size_t)i * 6 * D : nullptr;
  w.modf = w.mod_all ? w.mod_all + (size_t)p.depth * 6 * D : nullptr;
  const int nx = training ? p.depth + 1 : 2;
  w.X.resize(p.depth + 1);
  const size_t xs_ = p.x16 ? es : 4, gs_ = p.g16 ? es : 4;  // element size of the residual stream / of its gradient
  std::vector<char*> xs(nx);
  for (int i = 0; i < nx; ++i) xs[i] = take(BT * D * xs_);
  for (int i = 0; i <= p.depth; ++i) w.X[i] = xs[training ? i : (i & 1)];
  const int nb = training ? p.depth : 1;
  std::vector<BlockWS> bs(nb);
  for (int i = 0; i < nb; ++i) {
    BlockWS& b = bs[i];
    b.mean1 = (float*)take(BT * 4); b.rstd1 = (float*)take(BT * 4);
    b.mean2 = (float*)take(BT * 4); b.rstd2 = (float*)take(BT * 4);
    b.lse = (float*)take((size_t)B * p.H * p.T * 4);
    b.x_mid = take(BT * D * xs_);
    b.u1 = take(BT * D * es); b.qkv = take(BT * 3 * D * es); b.o = take(BT * D * es); b.y1 = take(BT * D * es);
    b.u2 = take(BT * D * es); b.hgrad = take(BT * M * es); b.h = take(BT * M * es); b.y2 = take(BT * D * es);
  }
  w.blk.resize(p.depth);
  for (int i = 0; i < p.depth; ++i) w.blk[i] = bs[training ? i : 0];
  w.meanf = (float*)take(BT * 4);
  w.rstdf = (float*)take(BT * 4);
  w.uf = take(BT * D * es);
  if (training) {
    w.zero_begin = base ? base + off : nullptr;
    const size_t z0 = off;
    w.dmod_base = (float*)take((size_t)B * p.ldmod() * 4);  // d modulation of every block + final layer; strides chosen per backward pass
    w.dsilu = (float*)take((size_t)B * D * 4);
    w.gxw = (float*)take((size_t)D * p.Pxpad * 4);
    w.dxpre = nullptr; w.gxmw = nullptr; w.gxmb = nullptr;
    if (p.mapper()) {
      w.gxmw = (float*)take((size_t)p.Pxpad * p.Ppad * 4);
      w.gxmb = (float*)take((size_t)p.Pxpad * 4);
    }
    w.gc0w = (float*)take((size_t)D * p.Kcpad * 4);
    w.gcmw = nullptr; w.gcmb = nullptr;
    if (p.cmapper()) {
      w.gcmw = (float*)take((size_t)p.Kcpad * p.Kcxpad * 4);
      w.gcmb = (float*)take((size_t)p.Kcpad * 4);
    }
    w.glin = (float*)take((size_t)p.Ppad * D * 4);
    w.glinb = (float*)take((size_t)p.Ppad * 4);
    w.G = (float*)take((size_t)p.T * D * 4);  // batch sum of d x0 (positional-table backward): accumulated with atomics
    w.zero_bytes = off - z0;
    w.dxA = take(BT * D * gs_);
    w.dxB = take(BT * D * gs_);
    w.delta = (float*)take((size_t)B * p.H * p.T * 4);
    for (int k = 0; k < 2; ++k) w.slab[k] = (float*)take(slab_bytes(p));
    w.dvp = take(BT * p.Ppad * es);
    if (p.mapper()) w.dxpre = take(BT * p.Pxpad * es);
    w.dcpre = p.cmapper() ? take((size_t)B * p.Kcpad * es) : nullptr;
    for (int k = 0; k < 2; ++k) {  // two sets, used alternately by consecutive blocks (lagged joins of the backward)
      w.dy[k] = take(BT * D * es);
      w.dy2[k] = take(BT * D * es);
      w.dhpre[k] = take(BT * M * es);
      w.dqkv[k] = take(BT * 3 * D * es);
    }
    w.du = take(BT * D * es);
    w.dof = take(BT * D * es);
    w.dmod_t = take((size_t)B * p.ldmod() * es);
    w.gtab = (float**)take(sizeof(PtrTable));
    w.dcond = take((size_t)B * D * es);
    w.dh_small = take((size_t)B * D * es);
    w.dh_small2 = take((size_t)B * D * es);
    w.dx0_t = take(BT * D * es);
  }
  w.total = off;
}

extern "C" size_t v4h_plan_workspace_bytes(const v4h_plan* p, int32_t B, int32_t training) {
  if (!p || B <= 0) return 0;
  WS w;
  layout(*p, B, training != 0, nullptr, w);
  return w.total;
}

// ------------------------------------------------------------------------------------------------ helpers
#define RUN(x)             \
  do {                     \
    int rc_ = (x);         \
    if (rc_) return rc_;   \
  } while (0)

struct Ctx {
  const v4h_plan& p;
  int B;
  const void* const* params;
  WS w;
  hipStream_t s;
  int BT() const { return B * p.T; }
  const float* pf(int i) const { return (const float*)params[i]; }
  const void* W(int i) const { return w.wop[i] ? (const void*)w.wop[i] : params[i]; }  // GEMM-operand view of a weight
};

static GemmArgs gargs(const void* P, int ldp, const void* Q, int ldq, int I, int J, int K) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.P = P; a.ldp = ldp; a.Q = Q; a.ldq = ldq; a.I = I; a.J = J; a.K = K;
  return a;
}

// Does the weight gradient of this shape go through split-K partial slabs + one reduce pass (else: f32 atomics / accumulation straight into dW)?
static bool wgrad_takes_slab(const v4h_plan& p, int I, int J, int K, int ldo) {
  const int sk = gemm_wgrad_splitk(p.mode, I, J, K);
  return sk > 1 && ldo == J && (size_t)sk * I * J * 4 <= slab_bytes(p) && (I * (long)J) % 4 == 0;
}
// dW[I][J] += dY^T X  (+ db[I] += column sums of dY).  `set` (only where wgrad_takes_slab): dW = dY^T X, whatever dW held before.
static int wgrad(const Ctx& c, const void* dY, int ld_dy, int I, const void* X, int ld_x, int J, int K, float* dW, int ldo, float* db, hipStream_t s = nullptr,
                 bool set = false) {
  GemmArgs a = gargs(dY, ld_dy, X, ld_x, I, J, K);
  a.e.out = dW; a.e.ldo = ldo; a.colsum = db;
  const int sk = gemm_wgrad_splitk(c.p.mode, I, J, K);
  hipStream_t st = s ? s : c.s;
  // Split-K partial sums: plain coalesced stores into a slab + one reduce pass instead of sk-fold f32 atomics (which run
  // at ~1.3 TB/s at the memory side): faster, and the weight gradient is bitwise reproducible.
  if (wgrad_takes_slab(c.p, I, J, K, ldo)) {
    // one scratch slab per queue (main: 0, side: 1): calls on one stream are ordered, calls on the two streams never share a slab
    V4H_CHECK_ARG(st == c.s || st == c.p.side, "wgrad: stream is neither the caller's nor the plan's side stream (no split-K scratch for it)");
    float* slab = c.w.slab[st == c.s ? 0 : 1];
    int nz = 1;
    RUN(gemm_wgrad_slab(c.p.mode, a, sk, slab, &nz, st));
    return slab_reduce(slab, nz, (long)I * J, dW, st, set);
  }
  V4H_CHECK_ARG(!set, "wgrad: overwrite requested for a shape that accumulates (I=%d J=%d K=%d)", I, J, K);
  return gemm_wgrad(c.p.mode, a, sk, st);
}

static int check_common(const v4h_plan* p, int B, const void* const* params, void* ws, size_t ws_bytes, bool training, const char* who) {
  V4H_CHECK_ARG(p != nullptr, "%s: null plan", who);
  V4H_CHECK_ARG(B > 0, "%s: empty batch (B=%d)", who, B);
  V4H_CHECK_ARG(params != nullptr && ws != nullptr, "%s: null parameter table / workspace", who);
  V4H_CHECK_ARG(((uintptr_t)ws % 256) == 0, "%s: workspace must be 256-byte aligned", who);
  const size_t need = v4h_plan_workspace_bytes(p, B, training);
  V4H_CHECK_ARG(ws_bytes >= need, "%s: workspace too small (%zu < %zu bytes)", who, ws_bytes, need);
  for (int i = 0; i < p->nparams(); ++i) V4H_CHECK_ARG(params[i] != nullptr && ((uintptr_t)params[i] % 16) == 0, "%s: parameter %d null or not 16-byte aligned", who, i);
  return V4H_OK;
}

// operand copies of the weights: cast to the mode type, awkward extents zero-padded, adaLN weights / biases concatenated (layout())
// stage of the pipelined update a parameter belongs to: 0 = everything the head of the forward needs (embedders, EVERY adaLN tensor - the modulation
// table of all blocks is one contraction at the head -, the final layer, fine-tuning mappers), 1 + i = the four Linears of block i
static int update_stage_of(const v4h_plan& p, int i) {
  if (i < P_BLOCK0 || i >= p.fin(0)) return 0;
  const int k = (i - P_BLOCK0) % B_COUNT;
  return k >= B_ADAW ? 0 : 1 + (i - P_BLOCK0) / B_COUNT;
}
static void operand_items(const Ctx& c, std::vector<CastPadItem>& items, int stage = -1) {  // stage -1: all
  const v4h_plan* p = &c.p;
  const WS& w = c.w;
  const int D = p->D;
  for (int i = 0; i < p->nparams(); ++i) {
    if (!w.wop[i]) continue;
    if (stage >= 0 && update_stage_of(*p, i) != stage) continue;
    int rp = p->rows[i], cp = p->cols[i];
    if (i == P_XW) cp = p->Pxpad;
    if (p->mapper() && i == p->xmw()) { rp = p->Pxpad; cp = p->Ppad; }
    if (i == P_C0W) cp = p->Kcpad;
    if (p->cmapper() && i == p->cmw()) { rp = p->Kcpad; cp = p->Kcxpad; }
    if (i == p->fin(F_LINW)) rp = p->Ppad;
    items.push_back(CastPadItem{c.pf(i), w.wop[i], p->rows[i], p->cols[i], rp, cp, 0});
  }
  if (stage > 0) return;
  items.push_back(CastPadItem{c.pf(p->fin(F_LINB)), w.linb_pad, 1, p->P, 1, p->Ppad, 1});
  for (int i = 0; i <= p->depth; ++i) {  // concatenated adaLN biases
    const bool last = i == p->depth;
    const int J = last ? 2 * D : 6 * D;
    items.push_back(CastPadItem{c.pf(last ? p->fin(F_ADAB) : p->blk(i, B_ADAB)), w.adaB + (size_t)i * 6 * D, 1, J, 1, J, 1});
  }
  if (p->mapper()) items.push_back(CastPadItem{c.pf(p->xmb()), w.xmb_pad, 1, p->Px, 1, p->Pxpad, 1});
  if (p->cmapper()) items.push_back(CastPadItem{c.pf(p->cmb()), w.cmb_pad, 1, p->Kc, 1, p->Kcpad, 1});
}

// Make the operand copies and the positional table of `params` in `ws` AHEAD of the next forward, on the plan's side stream: the 30 us cast of the 26 M
// parameters then runs beside whatever the caller enqueues on `stream` next (the update step's head: noise, trajectory, patch gather - small launches that
// use no weight), and the forward that follows - called with V4H_FWD_REUSE_OPERANDS on this very workspace, same B and training flag - waits for it just
// before its first weight-consuming kernel.  Ordered after everything enqueued on `stream` so far (the optimizer update).
extern "C" int32_t v4h_vit_prepare_operands(const v4h_plan* p, int32_t B, const void* const* params, void* ws, size_t ws_bytes, int32_t training, void* stream,
                                            const float* pos) {
  RUN(check_common(p, B, params, ws, ws_bytes, training != 0, "vit_prepare_operands"));
  V4H_CHECK_ARG(!p->mapped || pos != nullptr, "vit_prepare_operands: a plan made by v4h_plan_create_mapped needs d_pos");
  Ctx c{*p, B, params, WS(), (hipStream_t)stream};
  layout(*p, B, training != 0, (char*)ws, c.w);
  RUN(side_init(*p));
  hipStream_t st = c.s;
  if (g_overlap_wgrad) {
    RUN(side_wait_main(*p, c.s));
    st = p->side;
  }
  std::vector<CastPadItem> items;
  operand_items(c, items);
  RUN(cast_pad_many(p->mode, items.data(), (int)items.size(), st));
  if (pos) RUN(pos_embed_fwd_pos(c.pf(P_FREQS), pos, c.w.pe, p->T, p->D, st));
  else RUN(pos_embed_fwd(c.pf(P_FREQS), c.w.pe, p->pg, p->D, st));
  if (g_overlap_wgrad) {
    if (hipEventRecord(p->evOps, st) != hipSuccess) { v4h_set_error("vit_prepare_operands: cannot record"); return V4H_ERR_HIP; }
    p->ops_pending = true;
  }
  return V4H_OK;
}

// The optimizer update PIPELINED into the next step.  clip + AdamW over the flat parameter / gradient / moment buffers, stage by stage in the order the
// next forward consumes the weights, each stage followed by the operand copies of its tensors, all on the plan's side stream behind what `stream` holds
// so far (the gradient norm); an event per stage.  The next training forward on this workspace (V4H_FWD_REUSE_OPERANDS) waits for a stage's event only in
// front of its first use of that stage's weights: the 0.73 GB of AdamW traffic and the casts run beside the next step's head and first blocks - a
// bandwidth-bound kernel beside launch-latency-bound and matrix-bound ones - instead of between two steps.  `stream` is NOT ordered behind the update.
extern "C" int32_t v4h_vit_update_ahead(const v4h_plan* p, int32_t B, const void* const* params, float* flat_p, const float* flat_g, float* flat_m, float* flat_v,
                                        const int64_t* offsets, void* ws, size_t ws_bytes, const float* gnorm_sq, float max_norm, float lr0, float eta_min,
                                        int32_t t_max, float b1, float b2, float eps, float wd, float max_grad_norm, const int32_t* state_in, int32_t* state_out,
                                        int32_t* nonfinite, float* gnorm_out, void* stream, const float* pos) {
  RUN(check_common(p, B, params, ws, ws_bytes, true, "vit_update_ahead"));
  V4H_CHECK_ARG(flat_p && flat_g && flat_m && flat_v && offsets && state_in && state_out && state_in != state_out && t_max > 0, "vit_update_ahead: bad argument");
  V4H_CHECK_ARG(!p->mapper() && !p->cmapper(), "vit_update_ahead: networks with a fine-tuning mapper are updated by the caller's own optimizer");
  V4H_CHECK_ARG(!p->mapped || pos != nullptr, "vit_update_ahead: a plan made by v4h_plan_create_mapped needs d_pos");
  V4H_CHECK_ARG(p->depth + 1 <= 63, "vit_update_ahead: depth %d", p->depth);
  for (int i = 0; i < p->nparams(); ++i) {
    V4H_CHECK_ARG(params[i] == (const void*)(flat_p + offsets[i]), "vit_update_ahead: parameter %d is not the slice of the flat buffer its offset names", i);
    V4H_CHECK_ARG(offsets[i + 1] >= offsets[i] + (int64_t)p->rows[i] * (p->cols[i] ? p->cols[i] : 1), "vit_update_ahead: offsets of parameter %d overlap the next", i);
  }
  Ctx c{*p, B, params, WS(), (hipStream_t)stream};
  layout(*p, B, true, (char*)ws, c.w);
  RUN(side_init(*p));
  while ((int)p->evUp.size() < p->depth + 1) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { v4h_set_error("cannot create event"); return V4H_ERR_HIP; }
    p->evUp.push_back(e);
  }
  hipStream_t st = c.s;
  const bool side = g_overlap_wgrad;
  if (side) {
    RUN(side_wait_main(*p, c.s));
    st = p->side;
  }
  const AdamwHyper h{max_norm, lr0, eta_min, t_max, b1, b2, eps, wd, max_grad_norm, nullptr, 0.f};
  for (int stage = 0; stage <= p->depth; ++stage) {
    std::vector<long> lo, n;
    auto range = [&](int i0, int i1) { lo.push_back((long)offsets[i0]); n.push_back((long)(offsets[i1] - offsets[i0])); };
    if (stage == 0) {
      range(0, P_BLOCK0);
      for (int i = 0; i < p->depth; ++i) range(p->blk(i, B_ADAW), p->blk(i, B_ADAW) + 2);
      range(p->fin(0), p->nparams());
    } else {
      range(p->blk(stage - 1, 0), p->blk(stage - 1, B_ADAW));
    }
    for (size_t r0 = 0; r0 < lo.size(); r0 += ADAM_MAX_RANGES) {
      const int cnt = (int)std::min<size_t>(ADAM_MAX_RANGES, lo.size() - r0);
      RUN(adamw_step_ranges(flat_p, flat_g, flat_m, flat_v, lo.data() + r0, n.data() + r0, cnt, h, gnorm_sq, state_in, state_out, nonfinite, gnorm_out,
                            stage == 0 && r0 == 0, st));
    }
    std::vector<CastPadItem> items;
    operand_items(c, items, stage);
    if (!items.empty()) RUN(cast_pad_many(p->mode, items.data(), (int)items.size(), st));
    if (stage == 0) {
      if (pos) RUN(pos_embed_fwd_pos(c.pf(P_FREQS), pos, c.w.pe, p->T, p->D, st));
      else RUN(pos_embed_fwd(c.pf(P_FREQS), c.w.pe, p->pg, p->D, st));
    }
    if (side && hipEventRecord(p->evUp[stage], st) != hipSuccess) { v4h_set_error("vit_update_ahead: cannot record"); return V4H_ERR_HIP; }
  }
  p->upd_mask = side ? ((1ull << (p->depth + 1)) - 1) : 0;
  return V4H_OK;
}
// `stream` waits for everything the plan's side stream holds (a pipelined update, operand copies made ahead): call before the parameters, the moments or
// the workspace are touched from `stream` by anything but the next v4h_vit_forward.
// Gradient mode of the plan's backward passes: 0 (default) = gradients are ACCUMULATED into the caller's tensors (the caller zeroes them, or
// accumulates over several passes); 1 = every gradient tensor of the stages a call runs is WRITTEN by that call, whatever it held before.
extern "C" int32_t v4h_plan_set_gradient_mode(const v4h_plan* p, int32_t mode) {
  V4H_CHECK_ARG(p != nullptr && (mode == 0 || mode == 1), "plan_set_gradient_mode: null plan or mode %d not 0 (accumulate) / 1 (overwrite)", mode);
  p->grad_overwrite = mode == 1;
  return V4H_OK;
}
extern "C" int32_t v4h_plan_set_residual_storage(const v4h_plan* p, int32_t x_bf16, int32_t dx_bf16) {
  V4H_CHECK_ARG(p != nullptr, "plan_set_residual_storage: null plan");
  if (x_bf16 || dx_bf16) {
    V4H_CHECK_ARG(ln_resid16_supported(p->mode, p->D) && ln_resid_supported(p->D),
                  "plan_set_residual_storage: a bf16 residual stream needs V4H_MODE_BF16 and hidden_dim %d a multiple of 8 up to 512", p->D);
  }
  p->x16 = x_bf16 != 0;
  p->g16 = dx_bf16 != 0;
  return V4H_OK;
}
extern "C" int32_t v4h_plan_residual_storage(const v4h_plan* p) { return p ? (p->x16 ? 1 : 0) | (p->g16 ? 2 : 0) : 0; }
extern "C" int32_t v4h_plan_join(const v4h_plan* p, void* stream) {
  V4H_CHECK_ARG(p != nullptr, "plan_join: null plan");
  if (!p->side_ok) return V4H_OK;
  RUN(side_init(*p));
  RUN(main_wait_side(*p, (hipStream_t)stream));
  p->upd_mask = 0;
  p->ops_pending = false;
  return V4H_OK;
}

// ------------------------------------------------------------------------------------------------ forward
extern "C" int32_t v4h_vit_forward(const v4h_plan* p, int32_t B, const void* const* params, const float* x, const float* t, const float* cnd, float* out,
                                   void* ws, size_t ws_bytes, int32_t training, void* stream, const int32_t* pmap, const float* pos) {
  const int32_t flags = training;
  training = flags & 1;
  const bool reuse = (flags & V4H_FWD_REUSE_OPERANDS) != 0;  // operand copies + positional table of these parameters are already in this workspace
  const bool same_c = (flags & V4H_FWD_SAME_CONDITION) != 0;  // ... and so is the condition embedding of these conditions
  V4H_CHECK_ARG((flags & ~7) == 0, "vit_forward: unknown flag bits 0x%x", flags);
  V4H_CHECK_ARG(!same_c || (reuse && !training), "vit_forward: V4H_FWD_SAME_CONDITION needs V4H_FWD_REUSE_OPERANDS and an inference call");
  RUN(check_common(p, B, params, ws, ws_bytes, training != 0, "vit_forward"));
  RUN(check_geom(p, pmap, pos, "vit_forward"));
  V4H_CHECK_ARG(x && t && cnd && out, "vit_forward: null tensor");
  Ctx c{*p, B, params, WS(), (hipStream_t)stream};
  layout(*p, B, training != 0, (char*)ws, c.w);
  const WS& w = c.w;
  const Mode m = p->mode;
  const int BT = c.BT(), D = p->D, M = p->M, T = p->T;

  // 0. operand copies of the weights (cast to bf16 / zero-pad awkward extents), padded condition vector
  {
    std::vector<CastPadItem> items;
    if (!reuse) operand_items(c, items);
    if (same_c) {}
    else if (p->cmapper()) items.push_back(CastPadItem{cnd, w.cin, B, p->Kcx, B, p->Kcxpad, 0});
    else items.push_back(CastPadItem{cnd, w.cpad, B, p->Kc, B, p->Kcpad, 0});
    if (!items.empty()) RUN(cast_pad_many(m, items.data(), (int)items.size(), c.s));
  }
  // The token path (to_patches, positional table, x_embedder) and the conditioning path (t/c embedders, adaLN table) are
  // independent chains of small launch-latency-bound kernels until the first block: they run side by side.
  RUN(side_init(*p));
  // Only the c_embedder (two batch-row contractions, independent of everything else) forks to the side stream; the rest of the conditioning path
  // follows the token path on the MAIN stream.  With the whole-K kernel of v4h_gemm_small.h each of these launches is 5-7 us, less than one
  // cross-stream hand-over (an event record + wait is about 10 us of latency for the waiter: profiles/r03_step_timeline.txt), so the former layout -
  // whole conditioning chain on the side stream, joined before the first block - left the main stream idle for 80 us per forward.
  // (A third queue for the t_embedder was worth +0.3-0.9 % on one rank and serialised the whole pass as soon as a process group's communication
  // stream existed - 126 instead of 216 steps/s - and was removed in round 3.)
  if (p->upd_mask && !(reuse && training)) {  // a pipelined update is in flight and this is not the forward it was made for: simply join
    RUN(main_wait_side(*p, c.s));
    p->upd_mask = 0;
  }
  auto wait_update = [&](int stage) -> int {  // the weights of `stage` are about to be read: their AdamW + operand copies on the side stream must be done
    if (p->upd_mask & (1ull << stage)) {
      if (hipStreamWaitEvent(c.s, p->evUp[stage], 0) != hipSuccess) { v4h_set_error("vit_forward: cannot wait for the update of stage %d", stage); return V4H_ERR_HIP; }
      p->upd_mask &= ~(1ull << stage);
    }
    return V4H_OK;
  };
  const bool fork = g_overlap_wgrad && !same_c && !p->upd_mask;  // (behind a pipelined update the side stream is busy: the c_embedder stays on the main stream)
  hipStream_t cs = c.s;  // stream of the c_embedder
  if (fork) {
    RUN(side_wait_main(*p, c.s));
    cs = p->side;
  }
  // 1-3. to_patches, x_embedder + learnable positional embedding (nn/vit.py:193)
  char* patches = p->mapper() ? w.xpm : w.xp;  // (BT, Ppad) gathered voxels
  if (pmap) RUN(patchify_map(m, false, x, pmap, patches, B, p->V, T, p->P, p->Ppad, c.s));
  else RUN(patchify(m, x, patches, B, p->pg, p->P, p->Ppad, c.s));
  if (p->ops_pending) {  // operand copies requested ahead on the side stream (v4h_vit_prepare_operands): due now (the side stream is ordered behind them anyway)
    if (hipStreamWaitEvent(c.s, p->evOps, 0) != hipSuccess) { v4h_set_error("vit_forward: cannot wait for the operand copies"); return V4H_ERR_HIP; }
    p->ops_pending = false;
  }
  RUN(wait_update(0));
  if (p->mapper()) {  // fine-tuning embedding mapper: xp = silu(patches Wm^T + bm)   (experiment_finetuning.py:80-91)
    GemmArgs a = gargs(w.xpm, p->Ppad, c.W(p->xmw()), p->Ppad, BT, p->Pxpad, p->Ppad);
    a.e.out = w.xp; a.e.ldo = p->Pxpad; a.e.out2 = training ? w.xpre : nullptr; a.e.ldo2 = p->Pxpad; a.e.bias = (const float*)w.xmb_pad;
    RUN(gemm_fwd(m, EPI_SILU, a, c.s));
  }
  if (reuse) {}
  else if (pos) RUN(pos_embed_fwd_pos(c.pf(P_FREQS), pos, w.pe, T, D, c.s));
  else RUN(pos_embed_fwd(c.pf(P_FREQS), w.pe, p->pg, D, c.s));
  {
    GemmArgs a = gargs(w.xp, p->Pxpad, c.W(P_XW), p->Pxpad, BT, D, p->Pxpad);
    a.e.out = w.X[0]; a.e.ldo = D; a.e.bias = c.pf(P_XB); a.e.rowvec = w.pe; a.e.ld_rowvec = D; a.e.T = T; a.e.out_t = p->x16;
    RUN(gemm_fwd(m, EPI_EMBED, a, c.s));
  }
  // 4-8. c_embedder, t_embedder, cond = t_emb + c_emb, silu(cond) (nn/vit.py:197-199).  The condition term does not depend on t: a caller
  // that evaluates the network again for the same conditions (the ODE solver: 80 times per batch) says so and it is kept.
  {
    GemmArgs a;
    if (!same_c) {
      if (p->cmapper()) {  // fine-tuning condition mapper: c' = silu(c Wm^T + bm)   (experiment_finetuning.py:106-119)
        a = gargs(w.cin, p->Kcxpad, c.W(p->cmw()), p->Kcxpad, B, p->Kcpad, p->Kcxpad);
        a.e.out = w.cpad; a.e.ldo = p->Kcpad; a.e.out2 = training ? w.cpre : nullptr; a.e.ldo2 = p->Kcpad; a.e.bias = (const float*)w.cmb_pad;
        RUN(gemm_fwd(m, EPI_SILU, a, cs));
      }
      a = gargs(w.cpad, p->Kcpad, c.W(P_C0W), p->Kcpad, B, D, p->Kcpad);
      a.e.out = w.hc; a.e.ldo = D; a.e.out2 = w.hc_pre; a.e.ldo2 = D; a.e.bias = c.pf(P_C0B);
      RUN(gemm_fwd(m, EPI_SILU, a, cs));
      a = gargs(w.hc, D, c.W(P_C2W), D, B, D, D);
      a.e.out = w.cemb; a.e.ldo = D; a.e.out2 = w.silu_c; a.e.ldo2 = D; a.e.bias = c.pf(P_C2B);  // (silu_c: overwritten below)
      RUN(gemm_fwd(m, EPI_COND_SUM, a, cs));
    }
    RUN(timestep_embed(m, t, w.temb, B, p->F, c.s));
    a = gargs(w.temb, p->F, c.W(P_T0W), p->F, B, D, p->F);
    a.e.out = w.ht; a.e.ldo = D; a.e.out2 = w.ht_pre; a.e.ldo2 = D; a.e.bias = c.pf(P_T0B);
    RUN(gemm_fwd(m, EPI_SILU, a, c.s));
    if (fork) RUN(main_wait_side(*p, c.s));  // c_emb (long finished)
    a = gargs(w.ht, D, c.W(P_T2W), D, B, D, D);
    a.e.out = w.cond; a.e.ldo = D; a.e.out2 = w.silu_c; a.e.ldo2 = D; a.e.bias = c.pf(P_T2B); a.e.resid = w.cemb; a.e.ld_resid = D;
    RUN(gemm_fwd(m, EPI_COND_SUM, a, c.s));
  }
  // 9. every adaLN modulation of the step (nn/vit.py:323-330, 345-348)
  const int ldm = p->ldmod();
  if (w.adaW) {  // one contraction for the whole table: (B, D) x (ldmod, D)^T - seven launches of 18 workgroups each were pure launch latency
    GemmArgs a = gargs(w.silu_c, D, w.adaW, D, B, ldm, D);
    a.e.out = w.mod_all; a.e.ldo = ldm; a.e.bias = w.adaB;
    RUN(gemm_fwd(m, EPI_STORE_F32, a, c.s));
  } else {
    for (int i = 0; i <= p->depth; ++i) {
      const bool last = i == p->depth;
      const int J = last ? 2 * D : 6 * D;
      GemmArgs a = gargs(w.silu_c, D, c.W(last ? p->fin(F_ADAW) : p->blk(i, B_ADAW)), D, B, J, D);
      a.e.out = last ? w.modf : w.mod[i]; a.e.ldo = ldm; a.e.bias = c.pf(last ? p->fin(F_ADAB) : p->blk(i, B_ADAB));
      RUN(gemm_fwd(m, EPI_STORE_F32, a, c.s));
    }
  }
  // 10. DiT blocks (nn/vit.py:327-333)
  static const bool ln_resid = !(getenv("V4H_LN_RESID") && getenv("V4H_LN_RESID")[0] == '0');  // A/B hook
  const bool fuse_resid = ln_resid && ln_resid_supported(D);
  const bool x16 = p->x16;
  V4H_CHECK_ARG(!x16 || fuse_resid, "vit_forward: the bf16 residual stream exists only with the gated update fused into the LayerNorm kernels (V4H_LN_RESID=0 is set)");
  for (int i = 0; i < p->depth; ++i) {
    const BlockWS& b = w.blk[i];
    const float* mod = w.mod[i];
    RUN(wait_update(1 + i));
    // Gated residual updates (nn/vit.py:331-332).  Fused form (default): the branch contractions store y = branch output with a plain epilogue
    // and the NEXT LayerNorm kernel applies x += gate * y while it reads the row anyway - the f32 residual stream is then read and written once per
    // branch by a streaming kernel instead of by a contraction epilogue (16 us per call there, 8 here).  V4H_LN_RESID=0: the GATE_RESID epilogue.
    if (fuse_resid && i > 0) {
      const BlockWS& pb = w.blk[i - 1];
      RUN(ln_resid_modulate_fwd(m, pb.x_mid, pb.y2, w.mod[i - 1] + 5 * D, ldm, w.X[i], mod, mod + D, ldm, b.u1, b.mean1, b.rstd1, BT, T, D, c.s, x16));
    } else {
      RUN(ln_modulate_fwd(m, w.X[i], mod, mod + D, ldm, b.u1, b.mean1, b.rstd1, BT, T, D, c.s, x16));
    }
    GemmArgs a = gargs(b.u1, D, c.W(p->blk(i, B_QKVW)), D, BT, 3 * D, D);
    a.e.out = b.qkv; a.e.ldo = 3 * D; a.e.bias = c.pf(p->blk(i, B_QKVB));
    RUN(gemm_fwd(m, EPI_STORE, a, c.s));
    RUN(attention_fwd(m, b.qkv, b.o, b.lse, B, T, p->H, p->DH, c.s));
    a = gargs(b.o, D, c.W(p->blk(i, B_PROJW)), D, BT, D, D);
    if (fuse_resid) {
      a.e.out = b.y1; a.e.ldo = D; a.e.bias = c.pf(p->blk(i, B_PROJB));
      RUN(gemm_fwd(m, EPI_STORE, a, c.s));
      RUN(ln_resid_modulate_fwd(m, w.X[i], b.y1, mod + 2 * D, ldm, b.x_mid, mod + 3 * D, mod + 4 * D, ldm, b.u2, b.mean2, b.rstd2, BT, T, D, c.s, x16));
    } else {
      a.e.out = b.x_mid; a.e.ldo = D; a.e.out2 = training ? b.y1 : nullptr; a.e.ldo2 = D; a.e.bias = c.pf(p->blk(i, B_PROJB));
      a.e.rowvec = mod + 2 * D; a.e.ld_rowvec = ldm; a.e.T = T; a.e.resid = (const float*)w.X[i]; a.e.ld_resid = D;
      RUN(gemm_fwd(m, EPI_GATE_RESID, a, c.s));
      RUN(ln_modulate_fwd(m, b.x_mid, mod + 3 * D, mod + 4 * D, ldm, b.u2, b.mean2, b.rstd2, BT, T, D, c.s));
    }
    a = gargs(b.u2, D, c.W(p->blk(i, B_FC1W)), D, BT, M, D);
    a.e.out = training ? b.hgrad : nullptr; a.e.ldo = M; a.e.out2 = b.h; a.e.ldo2 = M;  /* hgrad = gelu_tanh'(fc1 output), h = gelu_tanh(fc1 output) */ a.e.bias = c.pf(p->blk(i, B_FC1B));
    RUN(gemm_fwd(m, EPI_GELU, a, c.s));
    a = gargs(b.h, M, c.W(p->blk(i, B_FC2W)), M, BT, D, M);
    if (fuse_resid) {
      a.e.out = b.y2; a.e.ldo = D; a.e.bias = c.pf(p->blk(i, B_FC2B));
      RUN(gemm_fwd(m, EPI_STORE, a, c.s));
    } else {
      a.e.out = w.X[i + 1]; a.e.ldo = D; a.e.out2 = training ? b.y2 : nullptr; a.e.ldo2 = D; a.e.bias = c.pf(p->blk(i, B_FC2B));
      a.e.rowvec = mod + 5 * D; a.e.ld_rowvec = ldm; a.e.T = T; a.e.resid = (const float*)b.x_mid; a.e.ld_resid = D;
      RUN(gemm_fwd(m, EPI_GATE_RESID, a, c.s));
    }
  }
  // 11. FinalLayer (nn/vit.py:347-351) with from_patches fused into the store
  if (fuse_resid && p->depth > 0) {
    const BlockWS& pb = w.blk[p->depth - 1];
    RUN(ln_resid_modulate_fwd(m, pb.x_mid, pb.y2, w.mod[p->depth - 1] + 5 * D, ldm, w.X[p->depth], w.modf, w.modf + D, ldm, w.uf, w.meanf, w.rstdf, BT, T, D, c.s, x16));
  } else {
    RUN(ln_modulate_fwd(m, w.X[p->depth], w.modf, w.modf + D, ldm, w.uf, w.meanf, w.rstdf, BT, T, D, c.s, x16));
  }
  {
    GemmArgs a = gargs(w.uf, D, c.W(p->fin(F_LINW)), D, BT, p->Ppad, D);
    a.e.out = out; a.e.bias = w.linb_pad; a.e.T = T; a.e.pg = p->pg; a.e.P = p->P; a.e.map = pmap; a.e.V = p->V;
    RUN(gemm_fwd(m, EPI_UNPATCH, a, c.s));
  }
  return V4H_OK;
}

// ------------------------------------------------------------------------------------------------ backward
// Backward of one adaLN Linear: weight / bias gradients and the contribution to d silu(cond).  Nothing on the main stream needs these
// before the embedder stage, so the three small launches (batch-row contractions: pure latency) run on the weight-gradient stream.
static int adaln_backward(const Ctx& c, const float* dmod, int J, int widx, int bidx, void* const* grads) {
  const WS& w = c.w;
  const int D = c.p.D, B = c.B;
  hipStream_t st = c.s;
  if (g_overlap_wgrad) {
    RUN(side_wait_main(c.p, c.s));  // dmod complete
    st = c.p.side;
  }
  CastPadItem it{dmod, w.dmod_t, B, J, B, J, 0};
  RUN(cast_pad_many(c.p.mode, &it, 1, st));
  RUN(wgrad(c, w.dmod_t, J, J, w.silu_c, D, D, B, (float*)grads[widx], D, (float*)grads[bidx], st));
  GemmArgs a = gargs(w.dmod_t, J, c.W(widx), D, B, D, J);  // M = B rows only: spread the long K over the chip instead
  a.e.out = w.dsilu; a.e.ldo = D;
  RUN(gemm_dgrad(c.p.mode, EPI_ATOMIC_F32, a, st, J / 96));
  if (g_overlap_wgrad) RUN(side_mark(c.p, S_ADA));  // d silu(cond) holds every contribution up to this block
  return V4H_OK;
}

// stage_events (optional, whole passes only): hipEvent_t per stage, recorded - on whichever stream finishes the stage's gradients - as soon as
// the gradient tensors of that stage are final, so the caller can start reducing them without the streams being joined at every stage.
static int backward_impl(const v4h_plan* p, int32_t B, const void* const* params, void* const* grads, const float* dout, void* ws, size_t ws_bytes,
                         int32_t stage_first, int32_t stage_last, void* stream, const int32_t* pmap, const float* pos, void* const* stage_events,
                         bool join = true) {
  RUN(check_common(p, B, params, ws, ws_bytes, true, "vit_backward"));
  RUN(check_geom(p, pmap, pos, "vit_backward"));
  V4H_CHECK_ARG(grads != nullptr, "vit_backward: null gradient table");
  for (int i = 0; i < p->nparams(); ++i) V4H_CHECK_ARG(grads[i] != nullptr && ((uintptr_t)grads[i] % 16) == 0, "vit_backward: gradient %d null or not 16-byte aligned", i);
  const int nst = p->depth + 2;
  V4H_CHECK_ARG(stage_first >= 0 && stage_last < nst && stage_first <= stage_last, "vit_backward: bad stage range [%d,%d] of %d", stage_first, stage_last, nst);
  V4H_CHECK_ARG(stage_first > 0 || dout != nullptr, "vit_backward: null output gradient");
  Ctx c{*p, B, params, WS(), (hipStream_t)stream};
  layout(*p, B, true, (char*)ws, c.w);
  const WS& w = c.w;
  const Mode m = p->mode;
  const int BT = c.BT(), D = p->D, M = p->M, T = p->T, depth = p->depth;
  RUN(side_init(*p));
  // (the side-stream marks of a pass issued as several calls WITHOUT a join in between - v4h_vit_backward_stage - stay live from call to call)
  if (stage_first == 0) p->mark_live = 0;
  // residual-stream gradient ping-pong: after stage s the live buffer is dx[(s+1)&1]... tracked explicitly below
  auto dxbuf = [&](int k) { return (k & 1) ? w.dxB : w.dxA; };
  // `forked(op)`: run the operator on the main stream and make the weight-gradient stream wait for exactly it (its completion signal, see arm_fork)
  auto forked = [&](auto&& op) -> int {
    if (!g_overlap_wgrad) return op();
    const ForkArm f = arm_fork(*p, c.s);
    const int rc = op();
    const int rc2 = complete_fork(*p, f, c.s, rc == V4H_OK);
    return rc != V4H_OK ? rc : rc2;
  };
  bool pre_forked = false;  // the side stream already waits for the kernel that wrote the next block's dy (the LayerNorm backward that ended the previous stage)
  // A backward pass issued as ONE call handles every adaLN Linear of the step together at the end (one cast, one grouped weight-gradient
  // contraction, one contraction for d silu(cond)) instead of three latency-bound launches per block: the d-modulation buffer is then one
  // (B, ldmod) table like the forward's.  Staged passes (gradient buckets reduced while later stages run) need each block's adaLN
  // gradients final at the end of its stage and keep the per-block launches, with per-block (B, 6 D) buffers in the same memory.
  const int ldm = p->ldmod();
  const bool whole_pass = stage_first == 0 && stage_last == depth + 1;
  auto stage_done = [&](int st, hipStream_t on) -> int {
    if (stage_events && stage_events[st] && hipEventRecord((hipEvent_t)stage_events[st], on) != hipSuccess) { v4h_set_error("vit_backward: cannot record the event of stage %d", st); return V4H_ERR_HIP; }
    return V4H_OK;
  };
  // (with stage events the caller reduces each block's gradients while the pass runs: every block's adaLN gradients must be final with its stage)
  const bool batch_ada = g_batch_adaln && whole_pass && !stage_events && w.adaW != nullptr && 3 * depth + 1 <= V4H_GEMM_MAX_GROUPS && D % 8 == 0;
  const int ldd = batch_ada ? ldm : 6 * D, lddf = batch_ada ? ldm : 2 * D;
  auto dmod = [&](int i) { return w.dmod_base + (batch_ada ? (size_t)i * 6 * D : (size_t)i * B * 6 * D); };
  float* const dmodf = dmod(depth);

  // Gradient mode "overwrite" (v4h_plan_set_gradient_mode): the four Linear weights of a block - 64 % of all gradient elements - are STORED by the
  // reduce pass of their split-K partials (no read-modify-write of zeros), and everything that is accumulated into (bias column sums, adaLN,
  // embedders, final layer) is zeroed here, by the one launch that also zeroes the workspace accumulators: the caller's 104 MB fill disappears.
  const bool overwrite = p->grad_overwrite;
  const bool set_qkv = overwrite && wgrad_takes_slab(*p, 3 * D, D, BT, D), set_proj = overwrite && wgrad_takes_slab(*p, D, D, BT, D);
  const bool set_fc1 = overwrite && wgrad_takes_slab(*p, M, D, BT, D), set_fc2 = overwrite && wgrad_takes_slab(*p, D, M, BT, M);
  if (overwrite) {
    std::vector<std::pair<float*, long>> zl;
    auto add = [&](int i) { zl.emplace_back((float*)grads[i], (long)p->rows[i] * (p->cols[i] ? p->cols[i] : 1)); };
    for (int st = stage_first; st <= stage_last; ++st) {
      if (st == 0) {
        zl.emplace_back((float*)w.zero_begin, (long)(w.zero_bytes / 4));
        for (int k = 0; k < F_COUNT; ++k)
          if (!(k == F_ADAW && batch_ada)) add(p->fin(k));
      } else if (st <= depth) {
        const int i = depth - st;
        for (int k = 0; k < B_COUNT; ++k) {
          const bool set = (k == B_QKVW && set_qkv) || (k == B_PROJW && set_proj) || (k == B_FC1W && set_fc1) || (k == B_FC2W && set_fc2) ||
                           (k == B_ADAW && batch_ada);  // (the grouped adaLN contraction of a whole pass stores its weight gradients)
          if (!set) add(p->blk(i, k));
        }
      } else {
        for (int i = 0; i < P_BLOCK0; ++i) add(i);
        for (int i = p->fin(0) + F_COUNT; i < p->nparams(); ++i) add(i);  // fine-tuning mappers
      }
    }
    RUN(zero_many(zl.data(), (int)zl.size(), c.s));
  }
  for (int st = stage_first; st <= stage_last; ++st) {
    if (st == 0) {
      if (!overwrite) {
        hipError_t e = hipMemsetAsync(w.zero_begin, 0, w.zero_bytes, c.s);
        if (e != hipSuccess) { v4h_set_error("vit_backward: memset failed: %s", hipGetErrorString(e)); return V4H_ERR_HIP; }
      }
      if (pmap) RUN(patchify_map(m, false, dout, pmap, w.dvp, B, p->V, T, p->P, p->Ppad, c.s));
      else RUN(patchify(m, dout, w.dvp, B, p->pg, p->P, p->Ppad, c.s));
      const bool ov0 = g_overlap_wgrad;
      hipStream_t s0 = ov0 ? p->side : c.s;
      if (ov0) RUN(side_wait_main(*p, c.s));  // dvp ready, accumulators zeroed
      RUN(wgrad(c, w.dvp, p->Ppad, p->Ppad, w.uf, D, D, BT, w.glin, D, w.glinb, s0));
      RUN(unpad_f32(w.glin, D, (float*)grads[p->fin(F_LINW)], p->P, D, s0));
      RUN(unpad_f32(w.glinb, 1, (float*)grads[p->fin(F_LINB)], p->P, 1, s0));
      GemmArgs a = gargs(w.dvp, p->Ppad, c.W(p->fin(F_LINW)), D, BT, D, p->Ppad);
      a.e.out = w.du; a.e.ldo = D;
      RUN(gemm_dgrad(m, EPI_STORE, a, c.s));
      LnBwdArgs l;
      memset(&l, 0, sizeof(l));
      l.du = w.du; l.x = w.X[depth]; l.mean = w.meanf; l.rstd = w.rstdf; l.scale = w.modf + D; l.ld_mod = p->ldmod();
      l.dx_out = dxbuf(0); l.dshift = dmodf; l.dscale = dmodf + D; l.ld_dmod = lddf;
      l.y = w.blk[depth - 1].y2; l.gate = w.mod[depth - 1] + 5 * D; l.ld_mod_gate = p->ldmod(); l.dy = w.dy[(depth - 1) & 1]; l.dgate = dmod(depth - 1) + 5 * D; l.ld_dgate = ldd;
      l.B = B; l.T = T; l.D = D; l.x16 = p->x16; l.g16 = p->g16;
      RUN(forked([&] { return ln_modulate_bwd(m, l, c.s); }));
      pre_forked = g_overlap_wgrad;
      if (!batch_ada) RUN(adaln_backward(c, dmodf, 2 * D, p->fin(F_ADAW), p->fin(F_ADAB), grads));
      RUN(stage_done(0, ov0 ? p->side : c.s));  // final-layer gradients: all on the weight-gradient stream
    } else if (st <= depth) {
      const int j = st - 1, i = depth - 1 - j;
      const BlockWS& b = w.blk[i];
      char* dx_in = dxbuf(2 * j);        // grad wrt X[i+1]
      char* dx_mid = dxbuf(2 * j + 1);   // grad wrt x_mid
      char* dx_out = dxbuf(2 * j + 2);   // grad wrt X[i] (same buffer as dx_in, which is dead by then)
      const bool ov = g_overlap_wgrad;
      hipStream_t ws_ = ov ? p->side : c.s;  // stream of the weight-gradient contractions
      // The four temporaries a block's weight gradients read exist twice; block i uses set i & 1.
      char *dy_i = w.dy[i & 1], *dy2_i = w.dy2[i & 1], *dh_i = w.dhpre[i & 1], *dq_i = w.dqkv[i & 1];
      // The main stream may not write a set before the weight gradients of block i+2, its previous readers, are done: ONE wait per block
      // (for the mark that block left behind its last weight gradient - two blocks old, long reached) instead of one per temporary.
      // [Tried: fewer forks - they cost a record packet each - by handing the side stream all four weight gradients at the block's end
      // (186.5 vs 189.2 steps/s) or two at a time (186.8 vs 193.6; again with the ring kernels of round 2: 219.7 vs 221.6): how finely the two
      // streams interleave matters more than the packets.]
      auto wg = [&](int k) -> int {
        switch (k) {
          case 0: return wgrad(c, dy_i, D, D, b.h, M, M, BT, (float*)grads[p->blk(i, B_FC2W)], M, (float*)grads[p->blk(i, B_FC2B)], ws_, set_fc2);
          case 1: return wgrad(c, dh_i, M, M, b.u2, D, D, BT, (float*)grads[p->blk(i, B_FC1W)], D, (float*)grads[p->blk(i, B_FC1B)], ws_, set_fc1);
          case 2: return wgrad(c, dy2_i, D, D, b.o, D, D, BT, (float*)grads[p->blk(i, B_PROJW)], D, (float*)grads[p->blk(i, B_PROJB)], ws_, set_proj);
          default: return wgrad(c, dq_i, 3 * D, 3 * D, b.u1, D, D, BT, (float*)grads[p->blk(i, B_QKVW)], D, (float*)grads[p->blk(i, B_QKVB)], ws_, set_qkv);
        }
      };
      auto fork_wgrad = [&](int k) -> int {  // one weight gradient on the side stream as soon as its dY exists
        if (ov) RUN(side_wait_main(*p, c.s));
        return wg(k);
      };
      if (ov) RUN(main_wait_mark(*p, S_BLK0 + (i & 1), c.s));
      // --- MLP branch (timm Mlp, nn/vit.py:317-322,332) ---
      if (pre_forked) RUN(wg(0));  // dy (and h) ready: the side stream waits for the kernel that wrote dy since the end of the previous stage
      else RUN(fork_wgrad(0));
      pre_forked = false;
      if (ov) RUN(side_mark(*p, S_FC2));
      GemmArgs a = gargs(dy_i, D, c.W(p->blk(i, B_FC2W)), M, BT, M, D);
      a.e.out = dh_i; a.e.ldo = M; a.e.aux = b.hgrad; a.e.ld_aux = M;
      RUN(forked([&] { return gemm_dgrad(m, EPI_DGELU, a, c.s); }));
      RUN(wg(1));  // dhpre ready
      a = gargs(dh_i, M, c.W(p->blk(i, B_FC1W)), D, BT, D, M);
      a.e.out = w.du; a.e.ldo = D;
      RUN(gemm_dgrad(m, EPI_STORE, a, c.s));
      LnBwdArgs l;
      memset(&l, 0, sizeof(l));
      l.du = w.du; l.x = b.x_mid; l.mean = b.mean2; l.rstd = b.rstd2; l.scale = w.mod[i] + 4 * D; l.ld_mod = p->ldmod();
      l.dx_in = dx_in; l.dx_out = dx_mid; l.dshift = dmod(i) + 3 * D; l.dscale = dmod(i) + 4 * D; l.ld_dmod = ldd;
      l.y = b.y1; l.gate = w.mod[i] + 2 * D; l.ld_mod_gate = p->ldmod(); l.dy = dy2_i; l.dgate = dmod(i) + 2 * D; l.ld_dgate = ldd;
      l.B = B; l.T = T; l.D = D; l.x16 = p->x16; l.g16 = p->g16;
      RUN(forked([&] { return ln_modulate_bwd(m, l, c.s); }));
      // --- attention branch (nn/vit.py:425-454,331) ---
      RUN(wg(2));  // dy2 ready
      a = gargs(dy2_i, D, c.W(p->blk(i, B_PROJW)), D, BT, D, D);
      a.e.out = w.dof; a.e.ldo = D;
      RUN(gemm_dgrad(m, EPI_STORE, a, c.s));
      RUN(forked([&] { return attention_bwd(m, b.qkv, b.o, w.dof, b.lse, w.delta, dq_i, B, T, p->H, p->DH, c.s); }));
      RUN(wg(3));  // dqkv ready
      if (ov) RUN(side_mark(*p, S_BLK0 + (i & 1)));
      a = gargs(dq_i, 3 * D, c.W(p->blk(i, B_QKVW)), D, BT, D, 3 * D);
      a.e.out = w.du; a.e.ldo = D;
      RUN(gemm_dgrad(m, EPI_STORE, a, c.s));
      // the LayerNorm backward below writes dy of block i-1 into the set block i+1 used: that block's fc2 weight gradient precedes this
      // block's in the side queue, and this block's was issued a whole block ago
      if (ov) RUN(main_wait_mark(*p, S_FC2, c.s));
      memset(&l, 0, sizeof(l));
      l.du = w.du; l.x = w.X[i]; l.mean = b.mean1; l.rstd = b.rstd1; l.scale = w.mod[i] + D; l.ld_mod = p->ldmod();
      l.dx_in = dx_mid; l.dshift = dmod(i); l.dscale = dmod(i) + D; l.ld_dmod = ldd;
      if (i > 0) {
        l.dx_out = dx_out;
        l.y = w.blk[i - 1].y2; l.gate = w.mod[i - 1] + 5 * D; l.ld_mod_gate = p->ldmod(); l.dy = w.dy[(i - 1) & 1]; l.dgate = dmod(i - 1) + 5 * D; l.ld_dgate = ldd;
      } else {
        l.dx_out_t = w.dx0_t;  // bottom of the stack: only the operand-typed copy is needed
      }
      l.B = B; l.T = T; l.D = D; l.x16 = p->x16; l.g16 = p->g16;
      if (i > 0) {  // writes dy of block i - 1: that block's first weight gradient waits for this kernel
        RUN(forked([&] { return ln_modulate_bwd(m, l, c.s); }));
        pre_forked = ov;
      } else {
        RUN(ln_modulate_bwd(m, l, c.s));
      }
      if (!batch_ada) RUN(adaln_backward(c, dmod(i), 6 * D, p->blk(i, B_ADAW), p->blk(i, B_ADAB), grads));
      RUN(stage_done(st, ws_));  // the block's weight gradients (and, unbatched, its adaLN gradients) are the last thing in the side queue
    } else {
      // --- embedders (nn/vit.py:76-82,193-199) ---
      // Three independent chains of small launches: x_embedder (+ mapper, positional table), c_embedder, t_embedder.  The first and the
      // last run on the side stream, the c_embedder chain on the main stream.
      const bool ov = g_overlap_wgrad;
      hipStream_t sx = ov ? p->side : c.s;
      if (batch_ada) {
        CastPadItem it{w.dmod_base, w.dmod_t, B, ldm, B, ldm, 0};
        RUN(cast_pad_many(m, &it, 1, c.s));
      }
      if (ov) RUN(side_wait_main(*p, c.s));  // d x0 (and the operand copy of d modulation) ready
      if (batch_ada) {
        // d silu(cond) = d mod . W_ada over K = ldmod, spread over the chip in K (M is only B rows): needed by the main stream next
        GemmArgs a = gargs(w.dmod_t, ldm, w.adaW, D, B, D, ldm);
        a.e.out = w.dsilu; a.e.ldo = D;
        RUN(gemm_dgrad(m, EPI_ATOMIC_F32, a, c.s, (ldm + 255) / 256));
        // every adaLN weight / bias gradient in one contraction: groups of 2 D rows (block i = groups 3i..3i+2, final layer = the last one)
        a = gargs(w.dmod_t, ldm, w.silu_c, D, ldm, D, B);
        a.e.ldo = D; a.e.group_rows = 2 * D;
        PtrTable tab;
        memset(&tab, 0, sizeof(tab));
        for (int i = 0; i <= depth; ++i) {
          float* gw = (float*)grads[i < depth ? p->blk(i, B_ADAW) : p->fin(F_ADAW)];
          float* gb = (float*)grads[i < depth ? p->blk(i, B_ADAB) : p->fin(F_ADAB)];
          for (int k = 0; k < (i < depth ? 3 : 1); ++k) {
            tab.p[3 * i + k] = gw + (size_t)k * 2 * D * D;
            tab.p[V4H_GEMM_MAX_GROUPS + 3 * i + k] = gb + (size_t)k * 2 * D;
          }
        }
        RUN(write_ptr_table(tab, w.gtab, sx));  // the gradient tensors may move between calls: the table is rewritten every time
        a.e.group_tab = w.gtab; a.e.out = tab.p[0]; a.colsum = tab.p[V4H_GEMM_MAX_GROUPS];
        a.e.store = overwrite ? 1 : 0;  // one writer per element (no K split): a plain store where the pass owns the gradient tensors
        RUN(gemm_wgrad(m, a, 1, sx));
      }
      RUN(wgrad(c, w.dx0_t, D, D, w.xp, p->Pxpad, p->Pxpad, BT, w.gxw, p->Pxpad, (float*)grads[P_XB], sx));
      RUN(unpad_f32(w.gxw, p->Pxpad, (float*)grads[P_XW], D, p->Px, sx));
      if (p->mapper()) {  // d pre = (d x0 . Wx) * silu'(pre) ; d Wm = d pre^T patches ; d bm = column sums
        GemmArgs a = gargs(w.dx0_t, D, c.W(P_XW), p->Pxpad, BT, p->Pxpad, D);
        a.e.out = w.dxpre; a.e.ldo = p->Pxpad; a.e.auxf = w.xpre; a.e.ld_auxf = p->Pxpad;
        RUN(gemm_dgrad(m, EPI_DSILU, a, sx));
        RUN(wgrad(c, w.dxpre, p->Pxpad, p->Pxpad, w.xpm, p->Ppad, p->Ppad, BT, w.gxmw, p->Ppad, w.gxmb, sx));
        RUN(unpad_f32(w.gxmw, p->Ppad, (float*)grads[p->xmw()], p->Px, p->P, sx));
        RUN(unpad_f32(w.gxmb, 1, (float*)grads[p->xmb()], p->Px, 1, sx));
      }
      if (ov && !batch_ada) RUN(main_wait_mark(*p, S_ADA, c.s));  // d silu(cond) has contributions from every adaLN backward on the side stream
      RUN(silu_bwd(m, w.dsilu, w.cond, w.dcond, B * D, c.s));
      // t_embedder: on the main stream with the c_embedder (the side stream already carries the grouped adaLN and the x_embedder weight gradients,
      // 110 of the stage's 250 us of launches: profiles/r03_step_timeline.txt)
      hipStream_t st2 = c.s;
      RUN(wgrad(c, w.dcond, D, D, w.ht, D, D, B, (float*)grads[P_T2W], D, (float*)grads[P_T2B], st2));
      GemmArgs a = gargs(w.dcond, D, c.W(P_T2W), D, B, D, D);
      a.e.out = w.dh_small2; a.e.ldo = D; a.e.auxf = w.ht_pre; a.e.ld_auxf = D;
      RUN(gemm_dgrad(m, EPI_DSILU, a, st2));
      RUN(wgrad(c, w.dh_small2, D, D, w.temb, p->F, p->F, B, (float*)grads[P_T0W], p->F, (float*)grads[P_T0B], st2));
      // c_embedder
      RUN(wgrad(c, w.dcond, D, D, w.hc, D, D, B, (float*)grads[P_C2W], D, (float*)grads[P_C2B]));
      a = gargs(w.dcond, D, c.W(P_C2W), D, B, D, D);
      a.e.out = w.dh_small; a.e.ldo = D; a.e.auxf = w.hc_pre; a.e.ld_auxf = D;
      RUN(gemm_dgrad(m, EPI_DSILU, a, c.s));
      RUN(wgrad(c, w.dh_small, D, D, w.cpad, p->Kcpad, p->Kcpad, B, w.gc0w, p->Kcpad, (float*)grads[P_C0B]));
      RUN(unpad_f32(w.gc0w, p->Kcpad, (float*)grads[P_C0W], D, p->Kc, c.s));
      if (p->cmapper()) {  // d pre = (d h . W_c0) * silu'(pre) ; d Wm = d pre^T c ; d bm = column sums
        a = gargs(w.dh_small, D, c.W(P_C0W), p->Kcpad, B, p->Kcpad, D);
        a.e.out = w.dcpre; a.e.ldo = p->Kcpad; a.e.auxf = w.cpre; a.e.ld_auxf = p->Kcpad;
        RUN(gemm_dgrad(m, EPI_DSILU, a, c.s));
        RUN(wgrad(c, w.dcpre, p->Kcpad, p->Kcpad, w.cin, p->Kcxpad, p->Kcxpad, B, w.gcmw, p->Kcxpad, w.gcmb));
        RUN(unpad_f32(w.gcmw, p->Kcxpad, (float*)grads[p->cmw()], p->Kc, p->Kcx, c.s));
        RUN(unpad_f32(w.gcmb, 1, (float*)grads[p->cmb()], p->Kc, 1, c.s));
      }
      if (pos) RUN(pos_embed_bwd_pos(m, w.dx0_t, c.pf(P_FREQS), pos, (float*)grads[P_FREQS], w.G, B, T, D, c.s));
      else RUN(pos_embed_bwd(m, w.dx0_t, c.pf(P_FREQS), (float*)grads[P_FREQS], w.G, B, p->pg, D, c.s));
    }
  }
  // Join: every gradient of the stages of this call is complete (in stream order) when the call returns, and no weight-gradient
  // kernel is left reading a temporary the next call may overwrite.
  if (g_overlap_wgrad && (join || stage_last == depth + 1)) RUN(main_wait_side(*p, c.s));
  if (stage_last == depth + 1) RUN(stage_done(depth + 1, c.s));
  return V4H_OK;
}

extern "C" int32_t v4h_vit_backward(const v4h_plan* p, int32_t B, const void* const* params, void* const* grads, const float* dout, void* ws, size_t ws_bytes,
                                    int32_t stage_first, int32_t stage_last, void* stream, const int32_t* pmap, const float* pos) {
  return backward_impl(p, B, params, grads, dout, ws, ws_bytes, stage_first, stage_last, stream, pmap, pos, nullptr);
}
// One stage of a pass that is issued stage by stage WITHOUT joining the library's two streams after every stage (the drop-in DDP route, round 5):
// `stage_event` (optional) is recorded - on whichever stream finishes them - when the gradients of THIS stage are final; the caller makes its stream wait for
// that event before it hands the stage's gradients on (one stage later, when the event has long been reached), instead of stalling the main stream for the
// weight-gradient stream's tail at every stage.  The last stage (embedders) always ends with the full join.  Stages must be issued in order 0 ... depth + 1.
extern "C" int32_t v4h_vit_backward_stage(const v4h_plan* p, int32_t B, const void* const* params, void* const* grads, const float* dout, void* ws, size_t ws_bytes,
                                          int32_t stage, void* stream, const int32_t* pmap, const float* pos, void* stage_event, int32_t join) {
  V4H_CHECK_ARG(p != nullptr && stage >= 0 && stage < p->depth + 2, "vit_backward_stage: null plan or bad stage %d", stage);
  std::vector<void*> evs;
  if (stage_event) {
    evs.assign(p->depth + 2, nullptr);
    evs[stage] = stage_event;
  }
  return backward_impl(p, B, params, grads, dout, ws, ws_bytes, stage, stage, stream, pmap, pos, stage_event ? evs.data() : nullptr, join != 0);
}
extern "C" int32_t v4h_vit_backward_events(const v4h_plan* p, int32_t B, const void* const* params, void* const* grads, const float* dout, void* ws,
                                           size_t ws_bytes, void* stream, const int32_t* pmap, const float* pos, void* const* stage_events) {
  V4H_CHECK_ARG(p != nullptr && stage_events != nullptr, "vit_backward_events: null plan or event table");
  for (int st = 0; st < p->depth + 2; ++st) V4H_CHECK_ARG(stage_events[st] != nullptr, "vit_backward_events: null event for stage %d", st);
  return backward_impl(p, B, params, grads, dout, ws, ws_bytes, 0, p->depth + 1, stream, pmap, pos, stage_events);
}

// ------------------------------------------------------------------------------------------------ CFM step pieces
extern "C" int32_t v4h_cfm_prepare(const float* x1, const float* x0, const float* t, float* xt, float* target, int32_t B, int64_t per, void* s) {
  V4H_CHECK_ARG(x1 && x0 && t && xt && target && B > 0 && per > 0 && per < (1LL << 31), "cfm_prepare: bad argument");
  return cfm_prepare(x1, x0, t, xt, target, B, (int)per, (hipStream_t)s);
}
// the two above for an update loop without fill launches: the trajectory kernel also zeroes the step's scalar accumulators, the loss kernel adds into one
extern "C" int32_t v4h_cfm_prepare_z(const float* x1, const float* x0, const float* t, float* xt, float* target, int32_t B, int64_t per, void* s, float* zero0,
                                     float* zero1) {
  V4H_CHECK_ARG(x1 && x0 && t && xt && target && B > 0 && per > 0 && per < (1LL << 31), "cfm_prepare_z: bad argument");
  return cfm_prepare(x1, x0, t, xt, target, B, (int)per, (hipStream_t)s, zero0, zero1);
}
extern "C" int32_t v4h_mse_loss_acc(const float* v, const float* target, float* loss, float* dv, int64_t n, void* s) {
  V4H_CHECK_ARG(v && target && loss && n > 0, "mse_loss_acc: bad argument");
  return mse_fwd_bwd(v, target, loss, dv, n, (hipStream_t)s, false);
}
extern "C" int32_t v4h_mse_loss(const float* v, const float* target, float* loss, float* dv, int64_t n, void* s) {
  V4H_CHECK_ARG(v && target && loss && n > 0, "mse_loss: bad argument");
  return mse_fwd_bwd(v, target, loss, dv, n, (hipStream_t)s);
}
extern "C" int32_t v4h_sq_norm_accum(const float* g, int64_t n, float* out, void* s) {
  V4H_CHECK_ARG(g && out && n > 0, "sq_norm_accum: bad argument");
  return sq_norm_accum(g, n, out, (hipStream_t)s);
}
extern "C" int32_t v4h_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq, float max_norm, float lr, float b1, float b2,
                                  float eps, float wd, int32_t step, void* s, int32_t* nonfinite) {
  V4H_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad argument");
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  return adamw_step(p, g, m, v, n, gnorm_sq, max_norm, lr, b1, b2, eps, wd, (float)bc1, (float)bc2, nonfinite, (hipStream_t)s);
}
extern "C" int32_t v4h_adamw_step_sched(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq, float max_norm, float lr0, float eta_min,
                                        int32_t t_max, float b1, float b2, float eps, float wd, const int32_t* state_in, int32_t* state_out, float max_grad_norm,
                                        void* s, int32_t* nonfinite, float* gnorm_out) {
  V4H_CHECK_ARG(p && g && m && v && n > 0 && state_in && state_out && t_max > 0, "adamw_step_sched: bad argument");
  V4H_CHECK_ARG(state_in != state_out, "adamw_step_sched: d_state_in and d_state_out must be distinct (every thread reads the one, one thread writes the other)");
  return adamw_step_sched(p, g, m, v, n, gnorm_sq, max_norm, lr0, eta_min, t_max, b1, b2, eps, wd, state_in, state_out, max_grad_norm, nonfinite, gnorm_out,
                          (hipStream_t)s);
}
extern "C" int32_t v4h_adamw_step_sched_ema(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq, float max_norm, float lr0, float eta_min,
                                            int32_t t_max, float b1, float b2, float eps, float wd, const int32_t* state_in, int32_t* state_out, float max_grad_norm,
                                            void* s, int32_t* nonfinite, float* gnorm_out, float* ema, float ema_decay) {
  V4H_CHECK_ARG(p && g && m && v && n > 0 && state_in && state_out && t_max > 0 && ema, "adamw_step_sched_ema: bad argument");
  V4H_CHECK_ARG(state_in != state_out, "adamw_step_sched_ema: d_state_in and d_state_out must be distinct");
  V4H_CHECK_ARG(ema_decay >= 0.0f && ema_decay <= 1.0f, "Decay must be between 0 and 1");  // torch_ema's own check and text
  return adamw_step_sched(p, g, m, v, n, gnorm_sq, max_norm, lr0, eta_min, t_max, b1, b2, eps, wd, state_in, state_out, max_grad_norm, nonfinite, gnorm_out,
                          (hipStream_t)s, ema, ema_decay);
}
extern "C" int32_t v4h_axpby(float* out, const float* a, const float* b, float alpha, float beta, int64_t n, void* s) {
  V4H_CHECK_ARG(out && a && b && n > 0, "axpby: bad argument");
  return axpby(out, a, b, alpha, beta, n, (hipStream_t)s);
}
extern "C" int32_t v4h_rk4_combine(float* y, const float* k1, const float* k2, const float* k3, const float* k4, float h, int64_t n, void* s) {
  V4H_CHECK_ARG(y && k1 && k2 && k3 && k4 && n > 0, "rk4_combine: bad argument");
  return rk4_combine(y, k1, k2, k3, k4, h, n, (hipStream_t)s);
}

extern "C" int32_t v4h_select_contraction_kernel(int32_t which) { return select_contraction_kernel(which); }
extern "C" int32_t v4h_selected_contraction_kernel(void) { return selected_contraction_kernel(); }

std::atomic<int> v4h_reserved_cus{0};
extern "C" int32_t v4h_reserve_compute_units(int32_t n) {
  V4H_CHECK_ARG(n >= 0 && n <= 64 && n % 8 == 0, "reserve_compute_units: %d is not a multiple of 8 in [0, 64]", n);
  v4h_reserved_cus.store(n, std::memory_order_relaxed);
  return V4H_OK;
}
extern "C" int32_t v4h_reserved_compute_units(void) { return v4h_reserved_cus.load(std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------------------ box calibration (bench.py)
extern "C" int32_t v4h_calib_mfma_loop(const void* rnd, float* sink, int32_t iters, int32_t blocks, void* s) {
  V4H_CHECK_ARG(rnd && sink && iters > 0 && blocks > 0 && ((uintptr_t)rnd % 16) == 0, "calib_mfma_loop: bad argument");
  return calib_mfma_loop(rnd, sink, iters, blocks, (hipStream_t)s);
}
extern "C" int32_t v4h_calib_copy(const void* src, void* dst, int64_t bytes, void* s) {
  V4H_CHECK_ARG(src && dst && bytes > 0 && bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "calib_copy: bad argument");
  return calib_copy(src, dst, (long)bytes, (hipStream_t)s);
}

// ------------------------------------------------------------------------------------------------ single operators
extern "C" int32_t v4h_op_gemm(int32_t mode, const void* P, int32_t ldp, int32_t pks, const void* Q, int32_t ldq, int32_t qks, const float* bias, void* out,
                               int32_t ldo, int32_t out_f32, int32_t I, int32_t J, int32_t K, int32_t splitk, float* colsum, void* s) {
  V4H_CHECK_ARG(mode == 0 || mode == 1, "op_gemm: bad mode");
  V4H_CHECK_ARG(P && Q && out, "op_gemm: null tensor");
  GemmArgs a = gargs(P, ldp, Q, ldq, I, J, K);
  a.e.out = out; a.e.ldo = ldo; a.e.bias = bias; a.colsum = colsum;
  if (!pks && !qks) return gemm_fwd((Mode)mode, out_f32 ? EPI_STORE_F32 : EPI_STORE, a, (hipStream_t)s);
  if (!pks && qks) return gemm_dgrad((Mode)mode, out_f32 ? EPI_STORE_F32 : EPI_STORE, a, (hipStream_t)s);
  if (pks && qks) {
    V4H_CHECK_ARG(out_f32 && !bias, "op_gemm: the wgrad form accumulates into an f32 output without bias");
    return gemm_wgrad((Mode)mode, a, splitk, (hipStream_t)s);
  }
  v4h_set_error("op_gemm: layout (P K-strided, Q K-contiguous) is not used on the path and not built");
  return V4H_ERR_UNSUPPORTED;
}
extern "C" int32_t v4h_op_gemm_gelu(int32_t mode, const void* x, int32_t ldx, const void* W, int32_t ldw, const float* bias, void* h, int32_t ldh, void* dh,
                                    int32_t ld_dh, int32_t I, int32_t J, int32_t K, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && x && W && h, "op_gemm_gelu: bad argument");
  GemmArgs a = gargs(x, ldx, W, ldw, I, J, K);
  a.e.out = dh; a.e.ldo = dh ? ld_dh : ldh; a.e.out2 = h; a.e.ldo2 = ldh; a.e.bias = bias;
  return gemm_fwd((Mode)mode, EPI_GELU, a, (hipStream_t)s);
}
extern "C" int32_t v4h_op_gemm_dgelu(int32_t mode, const void* dy, int32_t ld_dy, const void* W, int32_t ldw, const void* gelu_grad, int32_t ld_g, void* out,
                                     int32_t ldo, int32_t I, int32_t J, int32_t K, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && dy && W && gelu_grad && out, "op_gemm_dgelu: bad argument");
  GemmArgs a = gargs(dy, ld_dy, W, ldw, I, J, K);
  a.e.out = out; a.e.ldo = ldo; a.e.aux = gelu_grad; a.e.ld_aux = ld_g;
  return gemm_dgrad((Mode)mode, EPI_DGELU, a, (hipStream_t)s);
}
extern "C" int32_t v4h_op_gemm_wgrad_splitk(int32_t mode, int32_t I, int32_t J, int32_t K) {
  if ((mode != 0 && mode != 1) || I <= 0 || J <= 0 || K <= 0) return 1;
  return gemm_wgrad_splitk((Mode)mode, I, J, K);
}
extern "C" int32_t v4h_op_gemm_wgrad_slab(int32_t mode, const void* P, int32_t ldp, const void* Q, int32_t ldq, float* slab, float* out, int32_t I, int32_t J, int32_t K,
                                          int32_t splitk, float* colsum, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && P && Q && slab && out && splitk >= 1, "op_gemm_wgrad_slab: bad argument");
  V4H_CHECK_ARG(I > 0 && J > 0 && K > 0 && ldp >= I && ldq >= J, "op_gemm_wgrad_slab: empty problem or row strides shorter than the rows (I=%d J=%d K=%d ldp=%d ldq=%d)", I, J, K, ldp, ldq);
  // the reduction pass works on float4s and the partials of consecutive splits are I * J floats apart
  V4H_CHECK_ARG(((long)I * J) % 4 == 0, "op_gemm_wgrad_slab: I * J = %ld must be a multiple of 4", (long)I * J);
  V4H_CHECK_ARG(((uintptr_t)slab % 16) == 0 && ((uintptr_t)out % 16) == 0, "op_gemm_wgrad_slab: slab and out must be 16-byte aligned");
  GemmArgs a = gargs(P, ldp, Q, ldq, I, J, K);
  a.colsum = colsum;
  int nz = 0;
  if (int rc = gemm_wgrad_slab((Mode)mode, a, splitk, slab, &nz, (hipStream_t)s)) return rc;
  return slab_reduce(slab, nz, (long)I * J, out, (hipStream_t)s);
}
extern "C" int32_t v4h_op_attention_fwd(int32_t mode, const void* qkv, void* o, float* lse, int32_t B, int32_t T, int32_t H, int32_t dh, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && qkv && o && B > 0 && T > 0 && H > 0, "op_attention_fwd: bad argument");
  return attention_fwd((Mode)mode, qkv, o, lse, B, T, H, dh, (hipStream_t)s);
}
extern "C" int32_t v4h_op_attention_bwd(int32_t mode, const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int32_t B,
                                        int32_t T, int32_t H, int32_t dh, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && qkv && o && dout && lse && delta && dqkv && B > 0 && T > 0 && H > 0, "op_attention_bwd: bad argument");
  return attention_bwd((Mode)mode, qkv, o, dout, lse, delta, dqkv, B, T, H, dh, (hipStream_t)s);
}
extern "C" int32_t v4h_op_ln_modulate_fwd(int32_t mode, const float* x, const float* shift, const float* scale, int32_t ld_mod, void* u, float* mean, float* rstd,
                                          int32_t B, int32_t T, int32_t D, void* s) {
  V4H_CHECK_ARG((mode == 0 || mode == 1) && x && shift && scale && u && B > 0 && T > 0, "op_ln_modulate_fwd: bad argument");
  return ln_modulate_fwd((Mode)mode, x, shift, scale, ld_mod, u, mean, rstd, B * T, T, D, (hipStream_t)s);
}
extern "C" int32_t v4h_op_patchify(const v4h_plan* p, const float* vox, float* tok, int32_t B, void* s, const int32_t* pmap) {
  V4H_CHECK_ARG(p && vox && tok && B > 0, "op_patchify: bad argument");
  V4H_CHECK_ARG(p->mapped == (pmap != nullptr), "op_patchify: d_patch_map must be given exactly for mapped plans");
  if (pmap) return patchify_map(MODE_F32, true, vox, pmap, tok, B, p->V, p->T, p->P, p->P, (hipStream_t)s);
  return patchify(MODE_F32, vox, tok, B, p->pg, p->P, p->P, (hipStream_t)s);
}
extern "C" int32_t v4h_op_unpatchify(const v4h_plan* p, const float* tok, float* vox, int32_t B, void* s, const int32_t* pmap) {
  V4H_CHECK_ARG(p && vox && tok && B > 0, "op_unpatchify: bad argument");
  V4H_CHECK_ARG(p->mapped == (pmap != nullptr), "op_unpatchify: d_patch_map must be given exactly for mapped plans");
  if (pmap) return unpatchify_map_f32(tok, p->P, pmap, vox, B, p->V, p->T, p->P, (hipStream_t)s);
  return unpatchify_f32(tok, p->P, vox, B, p->pg, p->P, (hipStream_t)s);
}
extern "C" int32_t v4h_op_pos_embed(const v4h_plan* p, const float* freqs, float* pe, void* s, const float* pos) {
  V4H_CHECK_ARG(p && freqs && pe, "op_pos_embed: bad argument");
  V4H_CHECK_ARG(!p->mapped || pos != nullptr, "op_pos_embed: mapped plans need d_pos");
  if (pos) return pos_embed_fwd_pos(freqs, pos, pe, p->T, p->D, (hipStream_t)s);
  return pos_embed_fwd(freqs, pe, p->pg, p->D, (hipStream_t)s);
}
