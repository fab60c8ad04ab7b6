/*
 * jnroll.h — C ABI of libjnroll.so, the MI355X (gfx950) glimpse-rollout engine.
 *
 * The reference (jolibrain/jolineedle) has no FFI: its boundary for this path is the
 * Python operator API.  Each entry point below names the reference
 * interface it sits under (paths relative to /root/reference).  A reference-side
 * binding is a ctypes stub, shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; `*_dev` pointers are HIP device pointers owned by the caller
 *    (e.g. torch tensors' data_ptr()); `stream` is a hipStream_t passed as void*
 *    (NULL = the legacy default stream).
 *  - every function returns 0 on success or a negative JN_E* code; jn_last_error()
 *    gives the message of the calling thread's last failure.  No exceptions cross.
 *  - no hidden host synchronisation: only functions documented as "synchronises"
 *    wait for the device.
 *  - activations are fp32 (the reference is fp32 end to end); tensors exchanged at
 *    the boundary keep the reference's layouts (NCHW images/patches, [B,T] rows).
 */
#ifndef JNROLL_H
#define JNROLL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JN_ABI_VERSION 2

enum {
  JN_OK = 0,
  JN_EINVAL = -1,    /* bad argument / shape / state                     */
  JN_ENOMEM = -2,    /* device or host allocation failed                 */
  JN_EHIP = -3,      /* a HIP runtime call failed                        */
  JN_ENOTFOUND = -4, /* a required state-dict entry is missing           */
  JN_ESTATE = -5     /* call order violated (e.g. rollout before weights) */
};

/* Action-selection modes of jn_rollout (src/reinforce.py:73-90). */
enum { JN_MODE_GREEDY = 0, JN_MODE_SAMPLE = 1, JN_MODE_FORCED = 2 };

/* Which conv network a call addresses. */
enum { JN_NET_GPT_BACKBONE = 0, JN_NET_DETECTOR = 1 };

typedef struct jn_ctx jn_ctx;

/* Mirrors the fields of the reference's model_config / train_config that shape the
 * path (main.py:310-388; src/models/gpt.py:162-329).  Zero-initialise, set
 * struct_size = sizeof(jn_config). */
typedef struct jn_config {
  int32_t struct_size;
  int32_t device;               /* HIP device ordinal                                   */
  /* decision transformer: gpt.py:190-218 zoo entry already resolved */
  int32_t n_layer, n_head, n_embd;
  int32_t block_size;           /* max_seq_len T (tokens = T + 1 with the class token)  */
  int32_t n_actions;            /* 9 with --enable-stop else 8 (src/env/common.py:48-56) */
  int32_t patch_size;           /* P                                                    */
  int32_t use_pos_emb;          /* --use-positional-embedding                           */
  int32_t no_patch_emb;         /* --no-patch-embedding                                 */
  int32_t concat_emb;           /* --concat-embeddings                                  */
  int32_t decoder_pos_encoding; /* --decoder-pos-encoding                               */
  int32_t pos_emb_size;         /* rows of transformer.wpe (main.py:378)                */
  /* patch encoder `gpt_backbone` (gpt.py:261-264); width 0 = absent, the detector's
   * PAFPN encodes patches instead (gpt.py:376-380) */
  float gpt_bb_depth, gpt_bb_width;
  int32_t gpt_bb_depthwise;
  /* detector `yolox` (gpt.py:251-259); with_detector 0 = decision-only context */
  int32_t with_detector;
  float det_depth, det_width;
  int32_t det_depthwise;
  float det_conf_threshold;     /* --detector-conf-threshold                            */
  float det_nms_threshold;      /* yolox postprocess default 0.45                       */
  int32_t max_batch;            /* capacity B of env / rollout workspaces               */
  int32_t max_det_per_patch;    /* cap of kept boxes per patch (ragged output rows)     */
  int32_t act_dtype;            /* activation storage: 0 = fp32 (the reference's dtype, exact-fp32 MFMA),
                                 * 1 = bf16 (bf16 storage + bf16 MFMA, fp32 accumulate / statistics / weights) */
} jn_config;

/* One state-dict entry (names of SURVEY.md §5, e.g. "gpt_backbone.backbone.stem.conv.bn.weight"). */
typedef struct jn_tensor {
  const char* name;
  const void* data;   /* HOST pointer, contiguous                      */
  int32_t dtype;      /* 0 = float32, 1 = int64                        */
  int32_t ndim;
  int64_t shape[4];
} jn_tensor;

typedef struct jn_param_info {
  char name[160];
  int32_t dtype;      /* 0 = float32, 1 = int64                        */
  int32_t ndim;
  int64_t shape[4];
  int32_t is_buffer;  /* 1 = registered buffer (running stats, masks)  */
  int32_t used;       /* 0 = listed for state-dict compatibility only  */
} jn_param_info;

/* Outputs of one rollout: ReinforceTrainer.rollout's dict (src/reinforce.py:204-215).
 * All pointers are device memory sized for T = block_size steps; entries past
 * `n_steps` (read with jn_rollout_steps) are zero/false.  NULL = not wanted. */
typedef struct jn_rollout_out {
  float* rewards_dev;        /* [B, T]                                          */
  float* returns_dev;        /* [B, T]                                          */
  float* logprobs_dev;       /* [B, T]                                          */
  float* entropies_dev;      /* [B, T]                                          */
  uint8_t* masks_dev;        /* [B, T+1]                                        */
  uint8_t* logit_masks_dev;  /* [B, T]                                          */
  int64_t* positions_dev;    /* [B, T+1, 2] (y, x)                              */
  int64_t* actions_dev;      /* [B, T]   actions taken                          */
  float* logits_dev;         /* [B, T, n_actions] last-token logits per step    */
  float* final_emb_dev;      /* [B, T+1, n_embd] token embeddings (gpt.py:534)  */
  float* patches_dev;        /* [B, T+1, 3, P, P] or NULL                       */
  float* det_boxes_dev;      /* [B, T+1, max_det_per_patch, 7] or NULL          */
  int32_t* det_counts_dev;   /* [B, T+1] kept boxes per patch (0 = reference None) */
} jn_rollout_out;

/* ---- lifetime ------------------------------------------------------------------ */
int jn_abi_version(void);
const char* jn_last_error(void);
/* GPT.__init__ (src/models/gpt.py:162-329): builds the layer plan and workspaces. */
int jn_create(const jn_config* cfg, jn_ctx** out);
int jn_destroy(jn_ctx* ctx);

/* ---- weights: nn.Module.state_dict()/load_state_dict of GPT (main.py:532-584) ---- */
int jn_param_count(const jn_ctx* ctx);
int jn_param_info_at(const jn_ctx* ctx, int index, jn_param_info* out);
/* Uploads and pre-packs (BN folded for eval, Linear weights transposed).  Entries the
 * path does not use are ignored; a missing used entry -> JN_ENOTFOUND.  Synchronises. */
int jn_load_weights(jn_ctx* ctx, const jn_tensor* tensors, size_t n);

/* ---- environment: NeedleGeneralEnv (src/env/general_env.py) ----------------------- */
/* __init__ :15-82 + convert_bboxes_to_masks :360-379.  images [B,3,H,W] f32 stay owned
 * by the caller and must outlive the env; bboxes [B,nb,4] int64 xyxy (zero rows = pad). */
int jn_env_init(jn_ctx* ctx, const float* images_dev, const int64_t* bboxes_dev,
                int B, int H, int W, int nb, int max_ep_len, int stop_enabled, void* stream);
/* reset :144-170.  positions [B,2] (y,x) int64 or NULL = uniform draw from `seed`. */
int jn_env_reset(jn_ctx* ctx, const int64_t* positions_dev, uint64_t seed, void* stream);
/* step :172-233 + rewards :321-358 + terminated :235-246.  Outputs may be NULL. */
int jn_env_step(jn_ctx* ctx, const int64_t* actions_dev, float* rewards_dev,
                uint8_t* terminated_dev, uint8_t* truncated_dev, void* stream);
/* State views (device pointers into the context, valid until jn_env_init/jn_destroy):
 * what = 0 positions int64 [B,2]; 1 bbox_masks u8 [B,Gh,Gw]; 2 visited u8 [B,Gh,Gw];
 * 3 steps int32 [B]; 4 has_stopped u8 [B]. */
int jn_env_state(jn_ctx* ctx, int what, void** ptr_dev);
/* `patches` property :285-306 — bit-exact strided copy of the current patches
 * -> out [B,3,P,P] f32. */
int jn_env_patches(jn_ctx* ctx, float* out_dev, void* stream);
/* Stand-alone gather (no context state): out[b] = images[b,:,y*P:(y+1)*P, x*P:(x+1)*P]. */
int jn_gather_patches(const float* images_dev, const int64_t* positions_dev, float* out_dev,
                      int B, int C, int H, int W, int P, void* stream);
/* Trajectory form of the gather: out[n] = images[image_index[n], :, y_n*P:(y_n+1)*P, x_n*P:(x_n+1)*P] for N
 * (image, position) pairs — the patches NeedleSimpleEnv.generate_sample stacks one `get_patch` at a time
 * (src/env/simple_env.py:55-81, 472) and init_sample's detector patches (:417-419).  image_index[n] < 0 writes
 * a zero patch (a masked step of the zero-initialised sample, :380-384).  Positions must lie on the grid and
 * indices below n_images (the Python mirror asserts both, as get_patch does at :73-74). */
int jn_gather_patches_indexed(const float* images_dev, const int64_t* image_index_dev,
                              const int64_t* positions_dev, float* out_dev, int N, int n_images,
                              int C, int H, int W, int P, void* stream);

/* ---- detection augmentation ------------------------------------- */
/* Trainer.init_detection's on-device chain (src/trainer.py:176-186, applied at src/reinforce.py:332-333 and
 * src/supervised.py:855-861, 884-885) fused into one pass: RandomPlanckianJitter (per-patch red / blue gains,
 * clamp to [0,1]) -> RandomGrayscale -> RandomGaussianBlur 3x3 (reflect border) -> RandomPlasmaShadow (shade where a
 * value-noise fractal of the patch falls below `quantity`: kornia's diamond-square map restated as a counter-based
 * fractal, csrc/kernels_aug.hip) -> RandomGaussianNoise -> RandomMotionBlur 3x3 (zero border).  patches [N,3,P,P]
 * f32 -> out (must not alias); params [N,JN_AUG_NPARAM] f32 per patch = r_gain, b_gain, gray flag, Gaussian centre
 * weight, Gaussian side weight, noise std, motion kernel k[3][3] row-major, shade intensity, shade quantity,
 * roughness, fractal stretch, pad; an op a patch did not draw is encoded as the identity
 * (1, 1, 0, 1, 0, 0, delta, 0, ...).  noise_dev: optional [N,3,P,P]
 * standard-normal field (parity runs); NULL = counter-based generator seeded by `seed`. */
#define JN_AUG_NPARAM 20
int jn_augment_patches(const float* in_dev, float* out_dev, const float* params_dev, const float* noise_dev,
                       uint64_t seed, int N, int P, void* stream);

/* ---- networks ------------------------------------------------------------------- */
/* YOLOPAFPN.forward as called at src/models/gpt.py:375 / src/models/yolox.py:55 "with the
 * current mode of the model": train = 0 uses the BatchNorm running statistics, train != 0
 * the batch statistics of these N patches (and updates the running statistics, momentum
 * 0.03).  patches [N,3,P,P] f32 NCHW; fpn outputs NCHW f32 ([N,c,P/8,P/8], /16, /32), any
 * may be NULL. */
int jn_backbone_forward(jn_ctx* ctx, int net, const float* patches_dev, int N, int train,
                        float* fpn0_dev, float* fpn1_dev, float* fpn2_dev, void* stream);
/* Reads back a tensor the engine mutates (BatchNorm "<prefix>.bn.running_mean" / "running_var")
 * into host memory, so state_dict() of the owning module stays current.  Synchronises. */
int jn_read_tensor(jn_ctx* ctx, const char* name, float* host_out, size_t numel);
/* GPT.embed_patches (src/models/gpt.py:356-384): patches [N,3,P,P] -> [N, n_embd]. */
int jn_embed_patches(jn_ctx* ctx, const float* patches_dev, int N, float* out_dev, void* stream);
/* ---- training (loss.backward() / optimizer of src/reinforce.py:341-353) ---------------------- */
/* optimizer.zero_grad(): clears the flat gradient arena. */
int jn_zero_grad(jn_ctx* ctx, void* stream);
/* Backward of the most recent train-mode jn_backbone_forward(net, patches, N, train=1): g*_dev are
 * dL/d(fpn outputs), NCHW f32 (NULL = zero); parameter gradients (conv weights, BN weight/bias)
 * are ACCUMULATED into the gradient arena (read with jn_read_grad). */
int jn_backbone_backward(jn_ctx* ctx, int net, const float* patches_dev, int N, const float* g0_dev,
                         const float* g1_dev, const float* g2_dev, void* stream);
/* Options of one REINFORCE iteration (src/reinforce.py:217-265, 341). */
typedef struct jn_train_opts {
  int32_t struct_size;
  int32_t reward_norm;      /* config.reward_norm: advantages = (returns - mean) / (std + 1e-8)      */
  float ret_mean, ret_std;  /* last_return_mean / last_return_std of the previous optimiser window */
  float entropy_weight;     /* --entropy-weight                                                      */
  float loss_scale;         /* 1 / gradient_accumulation                                             */
} jn_train_opts;

/* One REINFORCE iteration minus the optimiser (src/reinforce.py:326-341): train-mode rollout
 * (batch-statistics BatchNorm per glimpse step, running statistics updated), loss + metrics, and
 * loss.backward() into the gradient arena (accumulating).  metrics_dev[8] = action_loss,
 * entropy_loss, loss, returns, episode_length, steps.  Synchronises once (to read the step count). */
int jn_reinforce_step(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev,
                      const int64_t* start_positions_dev, uint64_t seed, int stop_early,
                      const jn_train_opts* opts, const jn_rollout_out* out, float* metrics_dev, void* stream);
/* Autograd bridge (SURVEY.md 8b "Ownership": in training, rollout's logprobs / entropies carry a graph).  The reference
 * differentiates a loss built from the rollout dict (src/reinforce.py:326-341: rollout -> compute_metrics ->
 * (loss / ga).backward()).  jn_reinforce_forward is the train-mode rollout alone (batch-statistics BatchNorm, every
 * step's activations kept resident); jn_reinforce_backward is the backward of that rollout for GIVEN d loss / d logprobs
 * and d loss / d entropies ([B, T] f32, either may be NULL = zero): what a torch.autograd.Function around the rollout
 * receives.  Parameter gradients ACCUMULATE in the gradient arena.  jn_reinforce_backward synchronises once (step count). */
int jn_reinforce_forward(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev, const int64_t* start_positions_dev,
                         uint64_t seed, int stop_early, const jn_rollout_out* out, void* stream);
int jn_reinforce_backward(jn_ctx* ctx, const float* dlogprobs_dev, const float* dentropies_dev, void* stream);
/* The arena keeps tensors in kernel-friendly layouts (transposed Linear weights, tap-major conv weights, ...).  These
 * convert, ON THE DEVICE, between the arena and a caller-owned buffer of >= arena floats that holds every trainable
 * tensor in the reference's (PyTorch) layout at the SAME offset as in the arena (jn_arena_segment): the Python module's
 * param.data / param.grad are views of such buffers, so torch code (clip_grad_value_, an optimizer's state, a
 * state_dict) sees real tensors.  what = 0 parameters, 1 gradients, 2 / 3 AdamW exp_avg / exp_avg_sq (the optimiser
 * state of a checkpoint, torch.optim.AdamW.state_dict() layout per tensor).  export: arena -> buffer (accumulate != 0:
 * +=); import: buffer -> arena. */
int jn_arena_segment(jn_ctx* ctx, const char* name, size_t* off, size_t* numel);
int jn_export_arena(jn_ctx* ctx, int what, float* dst_dev, size_t numel, int accumulate, void* stream);
int jn_import_arena(jn_ctx* ctx, int what, const float* src_dev, size_t numel, void* stream);

/* --dropout (main.py:123-128 -> embd_pdrop = attn_pdrop = resid_pdrop, src/models/gpt.py:178-179): active in the
 * train-mode passes only (jn_reinforce_step / _forward, jn_supervised_step); eval entry points never drop.  The keep mask
 * of an element is a pure function of (seed + number of train-mode forwards since this call, agent, token, layer, site,
 * index) — Philox4x32-10, jn_device.h drop_scale — so the teacher-forced backward regenerates it and nothing is stored.
 * Deviation from the reference: a token's masks are drawn once, when the token is processed (KV cache),
 * whereas the reference re-draws the masks of the whole prefix at every glimpse step. */
int jn_set_dropout(jn_ctx* ctx, float p, uint64_t seed);
/* --freeze-image-processor (src/models/gpt.py:264-268: requires_grad = False on yolox.backbone.*): the optim_yolox
 * group (jn_optimizer_step_group(1)) then updates the detection head only. */
int jn_set_freeze(jn_ctx* ctx, int freeze_detector_backbone);
/* AdamW step counter of a parameter group (bias correction): read (set = 0) or restore (set != 0) — the "step" entry of
 * a torch optimizer state_dict, so that a resumed run continues where the checkpoint stopped. */
int jn_optimizer_steps(jn_ctx* ctx, int group, int* steps, int set);

/* One supervised (teacher-forced) step minus the optimiser: SupervisedTrainer.run body
 * (src/supervised.py:863-902) with the detector term off.  patches [B,T,3,P,P], current_actions /
 * next_actions [B,T] int64, classes [B] int64 (src/supervised.py:852, 866; NULL = class 0), positions [B,T,2] int64,
 * masks [B,T] u8 (1 = token, 0 = padding); B*T <=
 * max_batch.  GPT.forward runs on the full sequence in train mode (BatchNorm statistics over the B*T
 * patches), loss = CrossEntropy(weight[STOP] = stop_weight, reduction none) averaged over non-padding
 * tokens (:138-177); gradients ACCUMULATE in the arena.  logits_out_dev [B,T,n_actions] optional;
 * metrics_dev[4] = action_loss, action_accuracy, episode_length. */
int jn_supervised_step(jn_ctx* ctx, const float* patches_dev, const int64_t* current_actions_dev,
                       const int64_t* next_actions_dev, const int64_t* classes_dev, const int64_t* positions_dev,
                       const uint8_t* masks_dev, int B, int T, float stop_weight, float* logits_out_dev,
                       float* metrics_dev, void* stream);
/* Supervised autograd bridge — the two halves of jn_supervised_step around the CALLER's loss, so that the reference's
 * supervised loop runs unchanged (src/supervised.py:863-868: `action_logits, embeddings = model(patches, current_actions,
 * classes=classes, positions=positions)`, then :138-177 its cross-entropy, then :897 `loss.backward()`):
 * jn_supervised_forward = GPT.forward (src/models/gpt.py:481-534, full-sequence, train mode: BatchNorm statistics over
 * the B*T patches, running statistics updated, dropout) -> logits_out_dev [B,T,n_actions], final_emb_out_dev [B,T+1,C]
 * (optional); jn_supervised_backward = the backward of that forward for GIVEN d loss / d logits [B,T,n_actions],
 * gradients ACCUMULATE in the arena (the class token's gradient goes to row classes[b] of embed_class).  patches /
 * actions / classes / positions must stay alive until the backward; any pass over
 * the patch encoder in between makes the saved activations stale and the backward fails with JN_ESTATE. */
int jn_supervised_forward(jn_ctx* ctx, const float* patches_dev, const int64_t* current_actions_dev,
                          const int64_t* classes_dev, const int64_t* positions_dev, int B, int T, float* logits_out_dev,
                          float* final_emb_out_dev, void* stream);
int jn_supervised_backward(jn_ctx* ctx, const float* dlogits_dev, void* stream);
/* clip_grad_value_(clip_value) + AdamW (torch defaults) over the optim_gpt parameters
 * (src/reinforce.py:344-346, src/models/gpt.py:552-557); grad_scale multiplies the gradients first
 * (1/world_size after a SUM all-reduce). */
int jn_optimizer_step(jn_ctx* ctx, float lr, float weight_decay, float clip_value, float grad_scale, void* stream);
/* Sizes of the flat parameter/gradient arena (floats): whole arena, and its optim_gpt prefix. */
int jn_arena_info(jn_ctx* ctx, size_t* total_numel, size_t* optim_gpt_numel);
/* Use caller-owned device memory (>= arena floats, zeroed) as the gradient arena, e.g. a torch
 * tensor that is handed to one RCCL all-reduce per optimiser step. */
int jn_set_grad_arena(jn_ctx* ctx, float* grads_dev, size_t numel);
/* Current value of a trainable state-dict entry (reference layout) to host memory.  Synchronises. */
int jn_read_param(jn_ctx* ctx, const char* name, float* host_out, size_t numel);
/* param.grad of a trainable state-dict entry, in the reference's (PyTorch) layout, to host memory.
 * Synchronises. */
int jn_read_grad(jn_ctx* ctx, const char* name, float* host_out, size_t numel);

/* GPT.forward (src/models/gpt.py:481-534), eval mode.  patches [B,T,3,P,P] f32 (NULL with
 * no_patch_emb), actions [B,T] int64, positions [B,T,2] int64 (y,x; NULL unless use_pos_emb),
 * classes [B] int64 = row of embed_class behind every agent's class token (gpt.py:476-478; NULL = class 0, ids are
 * clamped to the table's 100 rows), prev_embeddings [B,Tp,C] or NULL.  Without prev_embeddings all T tokens are embedded
 * (1-D positions 0..T-1); with it only the last one is (1-D position 0, the reference's
 * recurrent quirk gpt.py:431-449) and appended.  L = prev ? Tp+1 : T+1.
 * Outputs: logits [B, L-1, n_actions], final_emb [B, L, n_embd] (either may be NULL). */
int jn_gpt_forward(jn_ctx* ctx, const float* patches_dev, const int64_t* actions_dev, const int64_t* classes_dev,
                   const int64_t* positions_dev, const float* prev_emb_dev, int B, int T, int Tp,
                   float* logits_dev, float* final_emb_dev, void* stream);
/* NeedleYOLOX.forward inference branch (src/models/yolox.py:24-57, 74-113): boxes
 * [N, max_det_per_patch, 7] (x1,y1,x2,y2,obj,cls,cls_id; clamped to [0,P-1]) + counts [N];
 * raw_dev optional [N, A, 6] decoded head output before postprocess. */
int jn_detect(jn_ctx* ctx, const float* patches_dev, int N, float* boxes_dev,
              int32_t* counts_dev, float* raw_dev, void* stream);

/* One training step of the detector minus the optimiser: NeedleYOLOX.forward(patches, targets) loss branch
 * (src/models/yolox.py:58-73) + loss.backward() (src/reinforce.py:336-341).  PAFPN + head forward with
 * batch-statistics BatchNorm on N patches, SimOTA assignment + IoU / objectness / class / L1 losses (YOLOX head,
 * use_l1 = True), backward; gradients are ACCUMULATED into the gradient arena (yolox.* parameters).
 * targets_dev: [N, nb, 5] float32 = (class id, x1, y1, x2, y2) in patch pixels, zero rows = padding
 * (NeedleGeneralEnv.get_detection_batch layout).  loss_scale multiplies the loss before backward
 * (1 / gradient_accumulation).  metrics_dev: float32[8] = total_loss, iou_loss (x5), conf_loss, cls_loss, l1_loss,
 * num_fg (foreground anchors per ground-truth box). */
int jn_detector_step(jn_ctx* ctx, const float* patches_dev, int N, const float* targets_dev, int nb,
                     float loss_scale, float* metrics_dev, void* stream);
/* Detector autograd bridge — NeedleYOLOX.forward(patches, targets) (src/models/yolox.py:24-91) as the two halves of
 * jn_detector_step around the CALLER's autograd, so that the reference's loop runs as written (src/reinforce.py:330-341:
 * `_, _, yolo_loss = yolox(patches_yolox, bboxes_yolox); loss += yolo_loss["total_loss"]; (loss / ga).backward()`; the
 * same in src/supervised.py:881-897).
 * jn_detector_forward: PAFPN + head in train mode on N <= max_batch patches (batch-statistics BatchNorm, running
 * statistics updated), SimOTA loss -> metrics_dev float32[8] as above.  The activations and d loss / d raw stay resident
 * as pass `pass` of `n_pass` (a detection batch above max_batch is fed in n_pass chunks, each with its own workspace
 * slot; pass 0 sizes the workspace for n_pass) until jn_detector_backward(pass).  Optional outputs of the rest of the
 * reference's forward: boxes_dev [N, max_det_per_patch, 7] + counts_dev [N] = the EVAL-mode head on the train-mode FPN
 * maps, postprocessed and clamped (yolox.py:74-91; both NULL: skipped), fpn{0,1,2}_dev = fpn_outs [N, C_i, H_i, W_i]
 * (activated, NCHW; NULL: skipped).  patches_dev must stay alive until the backward.
 * jn_detector_backward: the backward of pass `pass` for d L / d total_loss = dloss_dev[0] (device scalar, what torch hands
 * to the loss node; NULL = 1) times the host factor `scale` (the chunk's share N_pass / N_total of the batch mean);
 * yolox.* gradients ACCUMULATE in the arena.  One backward per forward; a later pass over the same slot or a second
 * backward fails with JN_ESTATE.  Independent of the REINFORCE / supervised bridges: either order of backward calls. */
int jn_detector_forward(jn_ctx* ctx, const float* patches_dev, int N, const float* targets_dev, int nb, int pass,
                        int n_pass, float* metrics_dev, float* boxes_dev, int32_t* counts_dev, float* fpn0_dev,
                        float* fpn1_dev, float* fpn2_dev, void* stream);
int jn_detector_backward(jn_ctx* ctx, int pass, const float* dloss_dev, float scale, void* stream);
/* AdamW + clip on one parameter group of the arena: group 0 = optim_gpt (everything but yolox.*,
 * src/models/gpt.py:552-557), group 1 = optim_yolox (yolox.*).  jn_optimizer_step == group 0. */
int jn_optimizer_step_group(jn_ctx* ctx, int group, float lr, float weight_decay, float clip_value,
                            float grad_scale, void* stream);

/* ---- the hot loop ----------------------------------------------------------------- */
/* ReinforceTrainer.rollout (src/reinforce.py:108-215) for the env set by jn_env_init:
 * reset (positions NULL = random from seed), then up to T steps of
 * patch-encode -> GPT decode (KV cache) -> action select -> env step [-> detect],
 * all enqueued on `stream` with no host synchronisation; the returns/logit_masks
 * epilogue (:186-202) runs on device too.  forced_actions [B,T] int64 for JN_MODE_FORCED.
 * stop_early != 0 reproduces the reference's `break` when every env is done (:181-184):
 * later steps become no-ops on device. */
int jn_rollout(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev,
               const int64_t* start_positions_dev, uint64_t seed, int do_detection,
               int stop_early, const jn_rollout_out* out, void* stream);
/* Number of steps S the last rollout executed (reference tensor width).  Synchronises
 * on `stream`. */
int jn_rollout_steps(jn_ctx* ctx, int* n_steps, void* stream);

/* Wall-clock helpers for bench.py: HIP-event time of the most recent jn_rollout on its
 * stream, split per kernel family.  what = 0 total ms, 1 backbone conv ms (forward, summed over the glimpse steps),
 * 2 conv-stack backward ms of the most recent jn_reinforce_step (embed_fpn + PAFPN backward).  Synchronises. */
int jn_last_timing(jn_ctx* ctx, int what, float* ms);
/* Enables HIP-event bracketing of the backbone conv section (costs two events/step). */
int jn_set_profiling(jn_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* JNROLL_H */
