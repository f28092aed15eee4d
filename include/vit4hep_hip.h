/* vit4hep_hip.h - C ABI of libvit4hep_hip.so: the MI355X (gfx950) implementation of the ViT-CFM hot path of
 * luigifvr/vit4hep.  Plain pointers and sizes only; every pointer named d_* is a DEVICE pointer (HBM), `stream`
 * is a hipStream_t passed as void*.  All entry points return 0 on success, non-zero on error
 * (v4h_last_error() gives the message); none of them synchronises, allocates device memory or touches the host
 * copy of any tensor, so a caller may capture them into a hipGraph (the network entry points fork onto a side stream owned by the
 * plan and join it again through events recorded on the caller's stream, which a capture follows; an inference forward is
 * captured and replayed in tests/test_hip_round2.py).
 *
 * Threads and devices.  A v4h_plan is NOT thread-safe: it owns a side stream, a ring of events and their cursor, all mutated by
 * every v4h_vit_forward / v4h_vit_backward* call, so calls on one plan must be serialised by the caller (one host thread per plan,
 * or a lock); distinct plans may be used from distinct threads once each contraction variant has been launched at least once (the
 * first launch of a variant sets its dynamic-LDS attribute without a lock) - warm up on one thread.  v4h_last_error() is
 * thread-local (the message of the calling thread's last failed call).  A plan binds to the device that is current at its first forward / backward call (side stream and
 * events live there); later calls with another device current are rejected.  Environment switches (V4H_*) are read once.
 *
 * The reference has no native boundary of its own (pure Python): each entry point below replaces
 * the PyTorch/timm/xformers/torchdiffeq calls of the cited reference lines (paths relative to the reference
 * repository root).  The Python host mirror of the reference classes that binds this ABI is
 * vit4hep_amd/nn/vit.py (class ViT) and vit4hep_amd/models (CFM, CaloChallengeCFM); see INTEGRATION.md.
 */
#ifndef VIT4HEP_HIP_H
#define VIT4HEP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V4H_ABI_VERSION 11

/* arithmetic mode of the contractions */
#define V4H_MODE_F32 0  /* exact f32 MFMA (v_mfma_f32_16x16x4_f32), f32 activations: parity mode (<= 1e-4 rel) */
#define V4H_MODE_BF16 1 /* bf16 MFMA (v_mfma_f32_16x16x32_bf16), f32 accumulate, bf16 activations: throughput mode */

/* Network + geometry description.  Mirrors ViT.__init__'s `param` mapping (nn/vit.py:52-73) and
 * CaloChallengeCFM.__init__ (experiments/calochallenge/calochallenge_cfm/model.py:9-38). */
typedef struct v4h_config {
  int32_t shape[3];       /* voxel grid (L, A, R)                 cfm_ds2_electrons.yaml:3  */
  int32_t patch_shape[3]; /* (p1, p2, p3)                         cfm_ds2_electrons.yaml:4  */
  int32_t in_channels;    /* must be 1                            cfm_ds2_electrons.yaml:2  */
  int32_t condition_dim;  /* 46                                   nn/vit.py:54              */
  int32_t hidden_dim;     /* 480                                  nn/vit.py:55              */
  int32_t depth;          /* 6                                    nn/vit.py:57              */
  int32_t num_heads;      /* 6                                    nn/vit.py:58              */
  int32_t mlp_hidden;     /* int(hidden_dim * mlp_ratio) = 1920   nn/vit.py:312             */
  int32_t freq_dim;       /* 256 TimestepEmbedder                 nn/vit.py:359             */
  int32_t mode;           /* V4H_MODE_*                                                     */
  int32_t x_embed_in;     /* 0, or: the x_embedder is Sequential(Linear(patch_dim -> x_embed_in), SiLU, Linear(x_embed_in -> hidden_dim)),
                             the embedding mapper of fine-tuning (experiments/calochallenge/calochallenge_cfm/experiment_finetuning.py:80-91).
                             Its two tensors (weight (x_embed_in, patch_dim), bias) come LAST in the parameter tables; x_embedder.{weight,bias} at
                             indices 1, 2 are then those of the inner Linear, (hidden_dim, x_embed_in). */
  int32_t c_embed_in;     /* 0, or: the c_embedder is Sequential(Linear(condition_dim -> c_embed_in), SiLU, <the backbone's c_embedder>), the
                             condition-embedding mapper of fine-tuning (experiment_finetuning.py:106-119).  condition_dim is then the width of the
                             conditions the caller passes, c_embed_in the input width of c_embedder.0 (indices 3, 4).  The mapper's two tensors
                             (weight (c_embed_in, condition_dim), bias) come LAST in the parameter tables, after those of the x mapper if both exist. */
} v4h_config;

typedef struct v4h_plan v4h_plan; /* host-side object: derived sizes and workspace offsets, no device state */

int32_t v4h_abi_version(void);
const char* v4h_last_error(void);

/* ---- plan ------------------------------------------------------------------------------------------------ */
int32_t v4h_plan_create(const v4h_config* cfg, v4h_plan** out);
/* General ("mapped") geometry: any voxel <-> patch-token layout, e.g. the multi-segment patching of CaloChallengeCFM_DS1
 * (calochallenge_cfm/model.py:97-173), CaloGANCFM (experiments/calogan/model.py:8-86), CaloHadCFM
 * (experiments/calohadronic/model.py:8-86).  cfg->shape / patch_shape are ignored; a sample is `voxels` consecutive f32
 * values, the network sees `tokens` tokens of `patch_dim` features.  Every forward / backward call then takes
 *   d_patch_map : int32 [tokens * patch_dim], voxel index (within the sample) of token n feature f, or -1 for none,
 *   d_pos       : f32 [3 * tokens] = pos_x | pos_y | pos_z, the buffers of ViT.create_meshgrid (nn/vit.py:137-154). */
int32_t v4h_plan_create_mapped(const v4h_config* cfg, int32_t tokens, int32_t patch_dim, int64_t voxels, v4h_plan** out);
void v4h_plan_destroy(v4h_plan* plan);
/* number of learnable tensors (= 11 + 10*depth + 4) and, for index i, its element count / rows / cols in the
 * reference's state_dict() order (nn/vit.py:76-132: pos_embed_freqs, x_embedder.{weight,bias},
 * c_embedder.{0,2}.{weight,bias}, t_embedder.mlp.{0,2}.{weight,bias}, blocks.i.{attn.qkv, attn.proj, mlp.fc1,
 * mlp.fc2, adaLN_modulation.1}.{weight,bias}, final_layer.{linear, adaLN_modulation.1}.{weight,bias}) */
int32_t v4h_plan_num_params(const v4h_plan* plan);
int32_t v4h_plan_param_shape(const v4h_plan* plan, int32_t index, int32_t* rows, int32_t* cols);
/* bytes of device workspace needed for batch size B; training != 0 keeps every activation the backward needs */
size_t v4h_plan_workspace_bytes(const v4h_plan* plan, int32_t B, int32_t training);

/* ---- network: CaloChallengeCFM.forward = to_patches -> ViT.forward -> from_patches
 *      (calochallenge_cfm/model.py:62-66, nn/vit.py:185-206) and its backward (autograd of the same) -------- */
/* d_params: host array of v4h_plan_num_params() device pointers to f32 tensors in the order above.
 * d_x (B,1,L,A,R) f32, d_t (B) f32, d_c (B,condition_dim) f32 -> d_out (B,1,L,A,R) f32.
 * `training`: bit 0 = keep every activation the backward needs (workspace sized with training = 1);
 *             bit 1 = V4H_FWD_REUSE_OPERANDS: the previous forward on this very workspace (same plan, B, bit 0) used the same,
 *             unchanged parameters - skip re-making their operand copies (bf16 casts, padded extents, positional table).
 *             The ODE sampler calls the network 80 times per batch with frozen weights (calochallenge_cfm/model.py:81-92). */
#define V4H_FWD_TRAINING 1
#define V4H_FWD_REUSE_OPERANDS 2
/* V4H_FWD_SAME_CONDITION (inference, together with REUSE_OPERANDS): d_c holds the same values as in the previous call on this workspace -
 * the c_embedder term of the conditioning (independent of t) is kept instead of recomputed.                                    */
#define V4H_FWD_SAME_CONDITION 4
int32_t v4h_vit_forward(const v4h_plan* plan, int32_t B, const void* const* d_params, const float* d_x, const float* d_t, const float* d_c,
                        float* d_out, void* d_workspace, size_t workspace_bytes, int32_t training, void* stream, const int32_t* d_patch_map,
                        const float* d_pos);
/* Operand copies AHEAD of the next forward.  v4h_vit_forward makes, per call, the contraction-operand copies of the parameters (bf16 casts, zero-padded
 * extents, concatenated adaLN tensors) and the positional table; a caller that knows the parameters of the NEXT forward already - the update step, right
 * after its optimizer kernel - calls this instead: the same work is enqueued on the plan's side stream, ordered after everything on `stream` so far, and the
 * next v4h_vit_forward on this workspace (same plan, B, training flag, parameters untouched in between) passes V4H_FWD_REUSE_OPERANDS and waits for the
 * copies just before its first weight-consuming kernel - so the cast of the 26 M parameters (30 us) runs beside the step's head (noise, trajectory,
 * patch gather of reference models/base_model.py:209-215) instead of in front of it.  Nothing is skipped: the same kernels, another queue. */
int32_t v4h_vit_prepare_operands(const v4h_plan* plan, int32_t B, const void* const* d_params, void* d_workspace, size_t workspace_bytes, int32_t training,
                                 void* stream, const float* d_pos);
/* The optimizer update PIPELINED into the next step (the fused update loop of vit4hep_amd/trainer.py; reference experiments/base_experiment.py:573-597:
 * clip_grad_norm_, AdamW, CosineAnnealingLR).  d_flat_p / g / m / v: one f32 buffer each holding every tensor of d_params at the element offsets
 * `offsets[i]` (host array of v4h_plan_num_params() + 1 entries; d_params[i] == d_flat_p + offsets[i]; the gaps between tensors are updated too and must
 * hold zeros).  The update of v4h_adamw_step_sched runs stage by stage in the order the next forward consumes the weights - embedders, every adaLN
 * tensor and the final layer first, then block 0 ... depth-1 -, each stage followed by the operand copies of its tensors (as v4h_vit_prepare_operands),
 * all on the plan's side stream behind what `stream` holds so far, with an event per stage.  The NEXT v4h_vit_forward on this workspace (training,
 * V4H_FWD_REUSE_OPERANDS, same B) waits for a stage's event only in front of its first use of that stage's weights, so the 0.73 GB of AdamW traffic runs
 * beside the next step's head and first blocks instead of between two steps; having waited for every stage, that forward leaves `stream` ordered behind
 * the whole update.  Until then `stream` is NOT ordered behind it: call v4h_plan_join before parameters, moments or workspace are touched otherwise.
 * (Any other forward on the plan joins first by itself.)  Same arithmetic as v4h_adamw_step_sched over the whole buffer: bit-identical results. */
int32_t v4h_vit_update_ahead(const v4h_plan* plan, int32_t B, const void* const* d_params, float* d_flat_p, const float* d_flat_g, float* d_flat_m,
                             float* d_flat_v, const int64_t* offsets, void* d_workspace, size_t workspace_bytes, const float* d_gnorm_sq, float max_norm,
                             float lr0, float eta_min, int32_t t_max, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                             const int32_t* d_state_in, int32_t* d_state_out, int32_t* d_nonfinite, float* d_gnorm_out, void* stream, const float* d_pos);
/* `stream` waits for everything the plan's side stream holds (a pipelined update, operand copies made ahead). */
int32_t v4h_plan_join(const v4h_plan* plan, void* stream);
/* Gradient mode of v4h_vit_backward / v4h_vit_backward_events on this plan.  0 (default): gradients are ACCUMULATED into d_grads - what autograd's
 * `.grad +=` and the reference's `optimizer.zero_grad(); loss.backward()` (experiments/base_experiment.py:559-560) need; the caller zeroes (or keeps
 * accumulating into) the tensors.  1: every gradient tensor of the stages a call runs is WRITTEN by that call, whatever it held before - the reduce pass
 * of the split-K partials of a block's four Linear weights stores instead of adding (no read-modify-write of zeros), the tensors that are accumulated
 * into with atomics are zeroed by the call itself, in the launch that zeroes its workspace accumulators anyway: an update loop that owns its gradient
 * buffer (vit4hep_amd/trainer.py) drops its zero fill of all gradients (104 MB per step at ds2).  Same values either way (bitwise for the weights). */
int32_t v4h_plan_set_gradient_mode(const v4h_plan* plan, int32_t mode);
/* Storage of the residual stream inside the workspace (ABI 11).  The reference keeps x of nn/vit.py:327-333 and its gradient in f32 because everything
 * there is f32; in V4H_MODE_BF16 every other activation of this library is already bf16 and the two f32 streams were 24 % of the step's HBM bytes for
 * no FLOPs.  x_bf16 != 0: the saved LayerNorm inputs (x before each block, x between a block's two branches) are stored as bf16; dx_bf16 != 0: the
 * gradient handed from one LayerNorm backward to the next is.  All arithmetic on them (gated update, row statistics, the sums of the backward) stays
 * f32 in registers.  Call BEFORE sizing the workspace (v4h_plan_workspace_bytes changes) and keep it fixed between a forward and its backward.
 * V4H_MODE_F32 plans and widths the 16-byte LayerNorm kernels do not serve (hidden_dim % 8 != 0 or > 512) refuse a non-zero request.  Default: 0, 0. */
int32_t v4h_plan_set_residual_storage(const v4h_plan* plan, int32_t x_bf16, int32_t dx_bf16);
/* bit 0: x is stored as bf16, bit 1: dx is */
int32_t v4h_plan_residual_storage(const v4h_plan* plan);
/* Backward of the forward that last filled d_workspace (training != 0).  d_dout (B,1,L,A,R) f32.
 * d_grads: host array of device pointers to f32 gradient tensors, same order/shapes as d_params; gradients are
 * ACCUMULATED into them (zero them first for a fresh gradient).  Stages allow overlap of the gradient all-reduce
 * with the remaining backward: stage 0 = final layer, stage 1+j = block depth-1-j, stage depth+1 = embedders
 * (x/t/c embedders + pos_embed_freqs).  Stages must be run in increasing order, each exactly once. */
int32_t v4h_vit_backward(const v4h_plan* plan, int32_t B, const void* const* d_params, void* const* d_grads, const float* d_dout, void* d_workspace,
                         size_t workspace_bytes, int32_t stage_first, int32_t stage_last, void* stream, const int32_t* d_patch_map, const float* d_pos);
int32_t v4h_vit_num_backward_stages(const v4h_plan* plan);
/* One stage of a backward pass issued stage by stage (stages in order 0 ... depth + 1, as v4h_vit_backward(stage, stage)) WITHOUT a join of the library's
 * two streams behind every stage (ABI 11; the route of the reference's DDP(model.net), experiments/base_experiment.py:161-167, whose bucket hooks fire per
 * autograd node): `stage_event` (hipEvent_t, optional) is recorded when this stage's gradient tensors are final, on whichever internal stream finishes
 * them; join != 0 additionally orders `stream` behind everything (what v4h_vit_backward does).  The last stage always joins.  A caller that passes join = 0
 * must make its stream wait for a stage's event before it reads that stage's gradients. */
int32_t v4h_vit_backward_stage(const v4h_plan* plan, int32_t B, const void* const* d_params, void* const* d_grads, const float* d_dout, void* d_workspace,
                               size_t workspace_bytes, int32_t stage, void* stream, const int32_t* d_patch_map, const float* d_pos, void* stage_event,
                               int32_t join);
/* The whole backward pass (stages 0 .. depth+1) as ONE call that still lets the caller overlap the gradient all-reduce: stage_events is a
 * host array of v4h_vit_num_backward_stages() hipEvent_t handles; event s is recorded - on whichever internal stream completes them - as
 * soon as the gradient tensors of stage s are final.  A communication stream that waits for event s may reduce that stage's gradients
 * while later stages still run (what DDP's bucket hooks do for the reference, experiments/base_experiment.py:161-167), and the two
 * internal streams are joined once per pass instead of once per stage.  `stream` itself is ordered after the complete pass on return. */
int32_t v4h_vit_backward_events(const v4h_plan* plan, int32_t B, const void* const* d_params, void* const* d_grads, const float* d_dout,
                                void* d_workspace, size_t workspace_bytes, void* stream, const int32_t* d_patch_map, const float* d_pos,
                                void* const* stage_events);

/* ---- energy-model velocity field: ParallelTransformer.forward (nn/cfm/transformer_cfm.py:12-119), forward only --------------
 * The network the reference samples the layer-energy ratios from before the shape model runs
 * (experiments/calochallenge/experiment.py:225-247; configs/model/cfm/cfm_ds{1,2,3}*_energy.yaml: `embeds: true`, one condition
 * token, d_model = 2 * dim_embedding = 128, 4 heads, 4 + 4 post-norm nn.Transformer layers, relu feed-forward 512).
 * Mirrors the `param` mapping of transformer_cfm.py:21-37. */
typedef struct v4h_energy_config {
  int32_t dims_in;            /* 45 (ds2/ds3), 5 / 7 (ds1)                 cfm_ds2_energy.yaml:15 */
  int32_t dims_c;             /* must be 1                                   cfm_ds2_energy.yaml:16 */
  int32_t dim_embedding;      /* 64 -> d_model 128                           transformer_cfm.py:45   */
  int32_t nhead;              /* 4  (head_dim 32)                                                    */
  int32_t num_encoder_layers; /* 4                                                                   */
  int32_t num_decoder_layers; /* 4                                                                   */
  int32_t dim_feedforward;    /* 512                                                                 */
  int32_t encode_t_dim;       /* 64, must equal dim_embedding                transformer_cfm.py:87-89 */
  int32_t mode;               /* V4H_MODE_*                                                          */
} v4h_energy_config;
typedef struct v4h_energy_plan v4h_energy_plan;
int32_t v4h_energy_plan_create(const v4h_energy_config* cfg, v4h_energy_plan** out);
void v4h_energy_plan_destroy(v4h_energy_plan* plan);
/* tensors in the module's named_parameters() order: time_embed.0.W, time_embed.1.{weight,bias}, x_embed.{weight,bias},
 * c_embed.{weight,bias}, pos_embed_x.weight, pos_embed_c.weight, layer.{weight,bias} (= layers.0), transformer.encoder.layers.i.
 * {self_attn.in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear1.*, linear2.*, norm1.*, norm2.*},
 * transformer.encoder.norm.*, transformer.decoder.layers.i.{self_attn.*, multihead_attn.*, linear1.*, linear2.*, norm1-3.*},
 * transformer.decoder.norm.*, layers.2.{weight,bias} */
int32_t v4h_energy_plan_num_params(const v4h_energy_plan* plan);
int32_t v4h_energy_plan_param_shape(const v4h_energy_plan* plan, int32_t index, int32_t* rows, int32_t* cols);
size_t v4h_energy_plan_workspace_bytes(const v4h_energy_plan* plan, int32_t B);
/* d_x (B, dims_in) f32, d_t (B) f32, d_c (B, 1) f32 -> d_out (B, dims_in) f32.  flags: V4H_FWD_REUSE_OPERANDS as above;
 * V4H_ENERGY_SAME_CONDITION: d_c holds the same values as in the previous call on this workspace - the encoder output and the
 * decoder's cross-attention terms (functions of the condition only) are reused: the ODE solver calls the network 80 times per
 * batch with one condition (models/base_model.py:231-242). */
#define V4H_ENERGY_SAME_CONDITION 4
#define V4H_ENERGY_COMPOSED 8 /* run the kernel-per-operator path even where the one-launch resident decoder applies (tests, A/B measurements) */
int32_t v4h_energy_forward(const v4h_energy_plan* plan, int32_t B, const void* const* d_params, const float* d_x, const float* d_t, const float* d_c,
                           float* d_out, void* d_workspace, size_t workspace_bytes, int32_t flags, void* stream);

/* ---- pre-/post-processing chain of the shape models, fused (experiments/calochallenge/transforms.py) ---------------------------
 * forward = NormalizeByElayer (331-397) -> ScaleTotalEnergy (184-202) -> CutValues (291-311, identity forward) ->
 * ExclusiveLogitTransform rescale=True (227-254, logit 11-18) -> GlobalStandardizeFromFile (21-64, stored statistics) -> LogEnergy
 * (149-164) -> ScaleEnergy (205-224) -> AddFeaturesToCond (130-146) -> Reshape (314-328), i.e. the `transforms` mapping of
 * configs/calochallenge/cfm/calochallenge_ds2.yaml:15-28; reverse = the same list backwards with rev=True (experiment.py:190-223). */
typedef struct v4h_chain_spec {
  int32_t n_layers;  /* NormalizeByElayer.n_layers (45)                                  */
  int64_t n_voxels;  /* layer_boundaries[-1] (6480 ds2, 40500 ds3, 368 ds1 photons)     */
  float eps;         /* NormalizeByElayer eps 1e-10                                      */
  float norm_cut;    /* NormalizeByElayer cut 0.0                                        */
  float factor;      /* ScaleTotalEnergy factor 0.35                                     */
  float cut;         /* CutValues cut 1e-7                                               */
  float delta;       /* ExclusiveLogitTransform delta 1e-6                               */
  float mean, std;   /* GlobalStandardizeFromFile statistics (means.npy / stds.npy)      */
  float alpha;       /* LogEnergy alpha 0                                                */
  float e_min, e_max;/* ScaleEnergy 6.907755 / 13.815510                                 */
} v4h_chain_spec;
/* d_layer_bounds: int32 [n_layers + 1] first voxel of every layer + n_voxels.
 * forward:  d_showers (B, n_voxels), d_energy (B) -> d_x (B, n_voxels) [= the Reshape'd network input], d_cond (B, n_layers + 1) = [u_0..u_{n-1} | energy]
 * reverse:  d_samples (B, n_voxels), d_cond (B, n_layers + 1) -> d_showers (B, n_voxels), d_energy (B) */
int32_t v4h_shape_preprocess(const v4h_chain_spec* spec, const int32_t* d_layer_bounds, const float* d_showers, const float* d_energy, float* d_x,
                             float* d_cond, int32_t B, void* stream);
int32_t v4h_shape_postprocess(const v4h_chain_spec* spec, const int32_t* d_layer_bounds, const float* d_samples, const float* d_cond, float* d_showers,
                              float* d_energy, int32_t B, void* stream);

/* ---- CFM step pieces ------------------------------------------------------------------------------------- */
/* linear_trajectory + target (models/trajectories.py:5-8, models/base_model.py:214): x_t = (1-t) x0 + t x1, target = x1 - x0 */
int32_t v4h_cfm_prepare(const float* d_x1, const float* d_x0, const float* d_t, float* d_xt, float* d_target, int32_t B, int64_t per_sample, void* stream);
/* loss = mean((v - target)^2) (models/base_model.py:217-218); d_dv (optional) = d loss / d v */
int32_t v4h_mse_loss(const float* d_v, const float* d_target, float* d_loss, float* d_dv, int64_t n, void* stream);
/* The same two for an update loop without fill launches (each is 5 us of serial stream time): v4h_cfm_prepare_z additionally sets *d_zero0 and *d_zero1
 * (optional device scalars: the step's loss and squared-norm accumulators) to 0; v4h_mse_loss_acc ADDS the loss into *d_loss instead of overwriting it. */
int32_t v4h_cfm_prepare_z(const float* d_x1, const float* d_x0, const float* d_t, float* d_xt, float* d_target, int32_t B, int64_t per_sample, void* stream,
                          float* d_zero0, float* d_zero1);
int32_t v4h_mse_loss_acc(const float* d_v, const float* d_target, float* d_loss, float* d_dv, int64_t n, void* stream);
/* sum of squares accumulated into d_out[0] (clip_grad_norm_, experiments/base_experiment.py:562-585) */
int32_t v4h_sq_norm_accum(const float* d_g, int64_t n, float* d_out, void* stream);
/* clip_grad_norm_(max_norm) + torch.optim.AdamW step on one flat tensor (experiments/base_experiment.py:573-592,
 * configs/training/default.yaml:5-10).  step >= 1 is the AdamW step count.  With d_gnorm_sq given (max_norm = +inf for "no clipping", which is
 * what the reference passes) a non-finite gradient norm leaves parameters and moments untouched and increments *d_nonfinite (optional, sticky,
 * device memory): the reference raises at that point (clip_grad_norm_(error_if_nonfinite=True), base_experiment.py:573-585), the caller of
 * this asynchronous form raises when it next looks at the flag.  While *d_nonfinite is non-zero EVERY later call skips (and counts) its update
 * as well, so the state the caller finds is exactly the one of the last finite step and no update was applied with a shifted step index; the caller
 * zeroes the counter to resume.  d_gnorm_sq == NULL: no clipping and no norm guard (a non-zero counter still holds updates back). */
int32_t v4h_adamw_step(float* d_p, const float* d_g, float* d_m, float* d_v, int64_t n, const float* d_gnorm_sq, float max_norm, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int32_t step, void* stream, int32_t* d_nonfinite);
/* The same update with the optimizer's step index and the learning-rate schedule position kept in DEVICE memory, so that an update the device decides to
 * skip advances neither - the reference's `_step` returns before optimizer.step() and scheduler.step() when the gradient norm exceeds
 * training.max_grad_norm after MIN_STEP_SKIP = 1000 iterations (experiments/base_experiment.py:31,586-591), and raises before them on a non-finite norm
 * (:573-585).  d_state_in / d_state_out: int32[4] each, DISTINCT buffers (every thread reads the one, one thread writes the other; the caller swaps them
 * between calls): [0] optimizer steps applied so far (torch.optim.AdamW's `step`), [1] scheduler steps so far (CosineAnnealingLR.last_epoch),
 * [2] updates skipped because of max_grad_norm, [3] reserved.  The update uses step = state[0] + 1 for the bias corrections and
 * lr = eta_min + (lr0 - eta_min) (1 + cos(pi state[1] / t_max)) / 2 (CosineAnnealingLR in closed form; configs/training/default.yaml:20-24), both
 * accurate to f32 rounding (1 - beta^step as -expm1(step log beta) with log beta taken in double on the host).  max_grad_norm: +inf = never skip (the caller passes +inf while its iteration index is <= MIN_STEP_SKIP).  max_norm, d_gnorm_sq
 * and d_nonfinite as in v4h_adamw_step.  d_gnorm_out (optional): receives sqrt(*d_gnorm_sq), the pre-clip norm clip_grad_norm_ returns. */
int32_t v4h_adamw_step_sched(float* d_p, const float* d_g, float* d_m, float* d_v, int64_t n, const float* d_gnorm_sq, float max_norm, float lr0, float eta_min,
                             int32_t t_max, float beta1, float beta2, float eps, float weight_decay, const int32_t* d_state_in, int32_t* d_state_out,
                             float max_grad_norm, void* stream, int32_t* d_nonfinite, float* d_gnorm_out);
/* The same with the shadow parameters of an exponential moving average updated in the pass (ABI 11; the reference keeps them in a
 * torch_ema.ExponentialMovingAverage and calls ema.update() right behind optimizer.step(), experiments/base_experiment.py:127-134,593-594; torch_ema is
 * un-vendored and unpinned there - its published update is restated): d_ema (f32, n elements, initialised by the caller with the parameters) becomes
 * d_ema - (1 - d) (d_ema - p_new) with d = min(ema_decay, (1 + k) / (10 + k)), k = the number of updates applied including this one (state[0] + 1:
 * torch_ema's num_updates warm-up).  A skipped update leaves the shadow untouched, as the reference's early return does. */
int32_t v4h_adamw_step_sched_ema(float* d_p, const float* d_g, float* d_m, float* d_v, int64_t n, const float* d_gnorm_sq, float max_norm, float lr0,
                                 float eta_min, int32_t t_max, float beta1, float beta2, float eps, float weight_decay, const int32_t* d_state_in,
                                 int32_t* d_state_out, float max_grad_norm, void* stream, int32_t* d_nonfinite, float* d_gnorm_out, float* d_ema,
                                 float ema_decay);
/* ODE solver vector updates for sample_batch (calochallenge_cfm/model.py:87-92; torchdiffeq fixed-grid solvers) */
int32_t v4h_axpby(float* d_out, const float* d_a, const float* d_b, float alpha, float beta, int64_t n, void* stream);
int32_t v4h_rk4_combine(float* d_y, const float* d_k1, const float* d_k2, const float* d_k3, const float* d_k4, float h, int64_t n, void* stream);

/* ---- single operators (unit parity tests; the network entry points above are built from these) ----------- */
/* out[i][j] = sum_k P[i][k] Q[j][k] (+ bias[j]); P/Q row-major with the given leading dims; *_kstrided != 0 means the
 * operand is stored [k][idx].  out is `mode`-typed unless out_f32 != 0.  nn.Linear forward/dgrad/wgrad. */
int32_t v4h_op_gemm(int32_t mode, const void* d_P, int32_t ldp, int32_t p_kstrided, const void* d_Q, int32_t ldq, int32_t q_kstrided, const float* d_bias,
                    void* d_out, int32_t ldo, int32_t out_f32, int32_t I, int32_t J, int32_t K, int32_t splitk, float* d_colsum, void* stream);
/* fc1 of the block's MLP with nn.GELU(approximate="tanh") fused into the contraction (timm Mlp, nn/vit.py:312-322): h[i][j] = gelu(sum_k x[i][k] W[j][k] + b[j])
 * (`mode`-typed, row stride ldh) and, when d_dh != NULL (training), the derivative gelu'(.) of the same pre-activation, which the backward multiplies in. */
int32_t v4h_op_gemm_gelu(int32_t mode, const void* d_x, int32_t ldx, const void* d_W, int32_t ldw, const float* d_bias, void* d_h, int32_t ldh, void* d_dh,
                         int32_t ld_dh, int32_t I, int32_t J, int32_t K, void* stream);
/* input gradient of fc2 times the saved GELU derivative: out[i][j] = (sum_k dy[i][k] W[k][j]) * gelu_grad[i][j]   (W stored [K][J] as nn.Linear keeps it) */
int32_t v4h_op_gemm_dgelu(int32_t mode, const void* d_dy, int32_t ld_dy, const void* d_W, int32_t ldw, const void* d_gelu_grad, int32_t ld_g, void* d_out,
                          int32_t ldo, int32_t I, int32_t J, int32_t K, void* stream);
/* Weight gradient as the backward pass computes it: out[i][j] += sum_k P[k][i] Q[k][j] (both operands token-major), split-K partials
 * into d_slab (f32, at least splitk * I * J elements) with plain stores, then one ordered reduction (bit-reproducible); optional
 * d_colsum[i] += sum_k P[k][i] (the bias gradient).  nn.Linear wgrad, reference nn/vit.py:416,420 + timm Mlp :317-322. */
int32_t v4h_op_gemm_wgrad_slab(int32_t mode, const void* d_P, int32_t ldp, const void* d_Q, int32_t ldq, float* d_slab, float* d_out, int32_t I, int32_t J,
                               int32_t K, int32_t splitk, float* d_colsum, void* stream);
/* The number of K splits the backward pass itself uses for a weight gradient of this shape (a function of the kernel it dispatches to and the 256 CUs). */
int32_t v4h_op_gemm_wgrad_splitk(int32_t mode, int32_t I, int32_t J, int32_t K);
/* softmax(q k^T / sqrt(dh)) v on token-major qkv (B*T, 3*H*dh) -> o (B*T, H*dh), lse (B,H,T)   nn/vit.py:425-451 */
int32_t v4h_op_attention_fwd(int32_t mode, const void* d_qkv, void* d_o, float* d_lse, int32_t B, int32_t T, int32_t H, int32_t dh, void* stream);
int32_t v4h_op_attention_bwd(int32_t mode, const void* d_qkv, const void* d_o, const void* d_do, const float* d_lse, float* d_delta, void* d_dqkv,
                             int32_t B, int32_t T, int32_t H, int32_t dh, void* stream);
/* LayerNorm(eps 1e-6, no affine) + modulate   nn/vit.py:309,457-458 ; shift/scale (B, ld_mod-strided) f32 */
int32_t v4h_op_ln_modulate_fwd(int32_t mode, const float* d_x, const float* d_shift, const float* d_scale, int32_t ld_mod, void* d_u, float* d_mean,
                               float* d_rstd, int32_t B, int32_t T, int32_t D, void* stream);
/* to_patches / from_patches   calochallenge_cfm/model.py:40-60 ; tokens are f32 (B*T, P) */
int32_t v4h_op_patchify(const v4h_plan* plan, const float* d_vox, float* d_tokens, int32_t B, void* stream, const int32_t* d_patch_map);
int32_t v4h_op_unpatchify(const v4h_plan* plan, const float* d_tokens, float* d_vox, int32_t B, void* stream, const int32_t* d_patch_map);
/* learnable_pos_embedding   nn/vit.py:156-162 -> (T, D) f32 */
int32_t v4h_op_pos_embed(const v4h_plan* plan, const float* d_freqs, float* d_pe, void* stream, const float* d_pos);

/* Which hand-written kernel serves the token-sized bf16 contractions (every Linear of a DiT block, nn/vit.py:317-322,416,420, and their dgrad /
   wgrad).  All choices compute the same contraction exactly (f32 accumulation in the same k order; they differ in where the bias enters the sum, i.e. in
   bf16 rounding): 0 = automatic (the measured winner per contraction class, the default), 1 = the 128 x 160 two-workgroup kernel everywhere,
   2 = the 256 x 160 ring kernel wherever the shape is eligible.  Process-global, read at launch time: set it while no call of another thread is being
   enqueued.  Used by the parity tests (bit-identical results across batch sizes need ONE kernel) and by A/B measurements.  Any other value is
   refused (V4H_ERR_ARG) and leaves the selection unchanged; ablation builds of the kernels exist only in libraries compiled with -DV4H_ABLATIONS
   and are not reachable through this header.  The environment variable V4H_GEMM2 (-1 / 0 / 8 = the same three choices) sets the initial value;
   other values are ignored with a message on stderr. */
int32_t v4h_select_contraction_kernel(int32_t which);
int32_t v4h_selected_contraction_kernel(void);

/* Leave n compute units (a multiple of 8 in [0, 64]; default 0) to a kernel of another library that runs beside the step - RCCL's ring kernels under
   DistributedDataParallel-style gradient all-reduce (experiments/base_experiment.py:161-167).  The contraction and single-chunk attention kernels
   launch ONE grid of persistent workgroups that each own a CU's whole LDS; with n reserved they launch 256 - n of them and the tile walk
   redistributes, instead of 256 of which those aimed at the CUs the communication kernel occupies wait for the rest of the grid to retire.  Changes
   scheduling only: every result is bit-identical for every n.  Process-global, read at launch time.  vit4hep_amd.parallel sets it when gradient
   collectives are enabled (profiles/r03_comm_interference.md). */
int32_t v4h_reserve_compute_units(int32_t n);
int32_t v4h_reserved_compute_units(void);

/* ---- box calibration (bench.py only; nothing of the reference corresponds to these) ----------------------------------------------------------
 * What THIS card at its clocks today does on a loop of nothing but bf16 MFMAs on random operands, and on a 16-byte-per-lane streaming copy: the
 * two figures travel in the bench line next to the step rate, so that rates measured on different boxes of the pool (2-4 % apart) can be compared.
 * v4h_calib_mfma_loop: `blocks` workgroups of 4 waves; every wave runs `iters` iterations of 16 v_mfma_f32_16x16x32_bf16 on operands taken from
 * d_rnd (bf16, at least blocks * 256 * 32 elements): 2 * 16 * 16 * 32 * 16 FLOP per wave and iteration.  d_sink: 64 floats (never written in practice).
 * v4h_calib_copy: dst[0 .. bytes) = src[0 .. bytes), bytes a multiple of 16. */
int32_t v4h_calib_mfma_loop(const void* d_rnd, float* d_sink, int32_t iters, int32_t blocks, void* stream);
int32_t v4h_calib_copy(const void* d_src, void* d_dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIT4HEP_HIP_H */
