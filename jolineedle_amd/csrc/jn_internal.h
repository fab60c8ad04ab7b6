// Internal structures of libjnroll.so (host side).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/jnroll.h"
#include "jn_kernels.h"

namespace jnr {

void set_error(const char* fmt, ...);

#define JN_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      jnr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return JN_EHIP;                                                                     \
    }                                                                                     \
  } while (0)

#define JN_CHECK(cond, code, ...)   \
  do {                              \
    if (!(cond)) {                  \
      jnr::set_error(__VA_ARGS__);   \
      return (code);                \
    }                               \
  } while (0)

// ---- activation buffers: NHWC fp32, per-image size known at plan time --------------
struct Buf {
  int H = 0, W = 0, C = 0;          // C = channel stride of a pixel (ld)
  size_t per_image() const { return (size_t)H * W * C; }
};

// A channel slice of a buffer.
struct View {
  int buf = -1;
  int H = 0, W = 0;
  int C = 0;      // channels in the view
  int coff = 0;   // first channel inside the buffer
};

enum OpKind {
  OP_STEM = 0,      // Focus + dense 3x3 (== 6x6 stride-2 on the image) + bias + SiLU
  OP_PW,            // 1x1 conv  (GEMM)   + bias + act (+ residual)
  OP_DW,            // depthwise 3x3 s1/s2 + bias + SiLU
  OP_CONV3,         // dense 3x3 s1/s2    + bias + SiLU       (non-depthwise models)
  OP_SPP,           // maxpool 5/9/13 of slice 0 into slices 1..3
  OP_UPSAMPLE,      // nearest x2 (raw values) into a slice
  OP_ADDACT,        // bottleneck shortcut: out = act(in) + act(res), materialised
  OP_PRED,          // YOLOXHead predictors of one level + decode (in = reg_feat, res = cls_feat)
};

struct Op {
  OpKind kind;
  View in, out;
  View res;                 // OP_ADDACT: the shortcut input (buf = -1: none)
  View alias;               // conv: second view that receives this layer's (scale, shift) (upsampled copy)
  int stride = 1;
  int act = ACT_SILU;
  int wslot = -1;           // index into Net::convs (packed weights)
  int level = 0, anchor0 = 0;   // OP_PRED
  bool acc_in = false;      // backward: add into g[in] (another consumer, later in forward order, wrote it first)
  bool acc_res = false;     // backward: same for g[res]
  std::string name;         // module prefix, e.g. "backbone.dark2.0.dconv"
};

// Packed (device) weights of one conv layer.
struct ConvW {
  std::string prefix;       // "<...>.conv" / "<...>.bn" live under this module prefix
  // Two 1x1 BaseConvs that read the same input (CSPLayer.conv2 and .conv1) run as ONE conv of cout channels:
  // rows [0, cout_first) belong to `prefix`, the rest to `prefix2`; their tensors are stored back to back.
  std::string prefix2;
  int cout_first = 0;
  int cin = 0, cout = 0, k = 1, groups = 1;
  bool has_bn = true;       // BaseConv; false = plain Conv2d with optional bias
  bool has_bias = false;    // plain Conv2d bias
  float* w_dev = nullptr;   // raw conv weight, layout depends on the op (see pack_conv)
  void* w_bf16 = nullptr;   // dense 3x3 in bf16 inference mode: the same [tap][N][K] layout pre-rounded to bf16
  float* b_dev = nullptr;   // [cout] bias of a BN-free Conv2d (else null)
  float *gamma_dev = nullptr, *beta_dev = nullptr;     // BN affine
  float *rmean_dev = nullptr, *rvar_dev = nullptr;     // BN running statistics (updated in train mode)
  int stat_off = 0;         // first channel of this layer in the per-slot stats / save arrays
};

struct Net {
  std::string prefix;       // "gpt_backbone." or "yolox.backbone."
  bool depthwise = false;
  float depth = 0, width = 0;
  int P = 0;
  std::vector<Buf> bufs;
  std::vector<Op> ops;
  std::vector<ConvW> convs;
  View fpn[3];              // pan_out2, pan_out1, pan_out0
  int n_backbone_ops = -1;  // ops before the detection head (-1: no head, all ops)
  size_t x3_lo = 0, x3_hi = 0;   // arena range of the 1x1 weights (multiples of 8 floats; hi == 0: not computed yet)
  bool x3_dirty = true;           // the bf16 planes of that range are older than the parameters (see mark_params_written)
  int n_anchors = 0, head_hid = 0;
  float* pred_w[3] = {nullptr, nullptr, nullptr};   // [6][hid] reg(4), obj, cls predictor rows
  float* pred_b[3] = {nullptr, nullptr, nullptr};   // [6]
  size_t per_image_floats = 0;
  std::vector<size_t> buf_off;    // per-image offset of each buffer (floats)
  std::vector<int> tab_off;       // first table channel of each buffer
  int tab_channels = 0;           // sum of buffer channels (table length)
  int stat_channels = 0;          // sum of BN channels
  // workspace slots: slot 0 = eval / single-step; slots 1..n = one per glimpse step when training
  int n_slots = 0;
  int act_dtype = 0;              // JN_F32 / JN_BF16 storage of the activation buffers
  char* act = nullptr;            // [n_slots][per_image_floats * max_batch] elements of act_dtype
  float* tab = nullptr;           // [n_slots][3][tab_channels]
  float* save = nullptr;          // [n_slots][2 * stat_channels]  (mean, invstd)
  double* stats = nullptr;        // [n_slots][2 * stat_channels]  (sum, sumsq)
  int g_slots = 0;                // slots of the three gradient-side buffers below (backward is step-batched)
  float* gact = nullptr;          // [g_slots] gradient buffers: mirror of one `act` slot each
  double* bred = nullptr;         // [g_slots][JN_NREP][2 * stat_channels] backward reductions (sum g_y, sum g_y * zhat)
  float* bconsts = nullptr;       // [g_slots][3 * stat_channels] per-channel backward constants
  bool eval_tab_dirty = true;     // slot-0 table must be rebuilt from the running statistics
  // deferred BatchNorm tables (ChanTab in jn_kernels.h): per table channel / per BatchNorm channel descriptors
  bool defer_ok = false, defer_built = false;
  int *td_src = nullptr, *td_goff = nullptr, *td_boff = nullptr; float* td_hw = nullptr;      // [tab_channels]
  float* fd_hw = nullptr; int *fd_goff = nullptr, *fd_boff = nullptr, *fd_t0 = nullptr, *fd_t1 = nullptr;   // [stat_channels]
  float **fd_rm = nullptr, **fd_rv = nullptr;
  std::vector<float> h_td_hw;     // host copies of the per-table-channel descriptors (td_hw: 0 where td_src < 0)
  std::vector<int> h_td_src, h_td_goff, h_td_boff;
};

struct ParamEntry {
  jn_param_info info;
};

// One trainable tensor inside the flat parameter arena.
struct ParamSeg {
  std::string name;          // the reference's state-dict name
  int kind = 0, d0 = 0, d1 = 0, d2 = 0;   // packing (api.hip: PackKind) and its dimensions
  size_t off = 0, numel = 0;
};

struct GptW {   // device, packed
  float *wte = nullptr, *wpe = nullptr, *embed_class = nullptr;
  float *proj_wt = nullptr, *proj_b = nullptr;          // project_concat: Wt [n_in][C]
  float *pos1d = nullptr;                                // [T+1][C] 1-D sinusoid table
  float *pos2d_col = nullptr, *pos2d_row = nullptr;      // [256][ch2] tables (column / row halves)
  float *efpn_w = nullptr;                               // embed_fpn.0 weight [C][cin]
  float *efpn_lin_wt = nullptr, *efpn_lin_b = nullptr;   // embed_fpn.3: Wt [(h*w)*C + c][C]
  float *head_wt = nullptr;                              // [C][nA]
  float *lnf_w = nullptr, *lnf_b = nullptr;
  struct Layer {
    float *ln1_w, *ln1_b, *qkv_wt, *qkv_b, *proj_wt, *proj_b, *ln2_w, *ln2_b, *fc_wt, *fc_b, *fc2_wt, *fc2_b;
  };
  std::vector<Layer> layers;
};

struct EnvState {
  bool ready = false;
  const float* images = nullptr;
  int B = 0, H = 0, W = 0, nb = 0, Gh = 0, Gw = 0, T = 0, stop = 0;
  int64_t* positions = nullptr;   // [B,2]
  uint8_t* bbox_masks = nullptr;  // [B,Gh,Gw]
  uint8_t* visited = nullptr;     // [B,Gh,Gw]
  int32_t* steps = nullptr;       // [B]
  uint8_t* has_stopped = nullptr; // [B]
  int32_t* n_bbox_tiles = nullptr;// [B] sum(bbox_masks)
};

}  // namespace jnr

struct jn_ctx {
  jn_config cfg;
  std::vector<jnr::ParamEntry> params_tab;
  const std::vector<jnr::ParamEntry>& params_table() const { return params_tab; }
  // flat trainable-parameter arena + gradient / AdamW mirrors
  float* params = nullptr; float* grads = nullptr; float* adam_m = nullptr; float* adam_v = nullptr;
  uint16_t* params_x3 = nullptr;   // 3 bf16 per arena float: the 1x1 weights split for pw_x3_kernel (refreshed at the start of every fp32 pass)
  uint16_t* params_x3t = nullptr;  // the TRANSPOSED 1x1 weights of the wide layers as three bf16 planes (data gradient, split per backward)
  size_t arena_size = 0, arena_used = 0, gpt_arena_end = 0;   // [0, gpt_arena_end) = optim_gpt parameters
  int adam_step = 0, adam_step_yolox = 0;
  bool freeze_det_backbone = false;   // --freeze-image-processor: yolox.backbone.* keep their values (src/models/gpt.py:264-268)
  size_t det_head_begin = 0;          // arena offset of the first yolox.head.* tensor
  jnr::GptLayerPtrs* g_layers_dev = nullptr;
  float* efpn_train = nullptr;    // [T][B][h*w*C] embed_fpn.0 activations of every glimpse step
  float* tok_emb_train = nullptr; // [B][T][C] patch embeddings of every glimpse step
  float* d_tok_emb = nullptr;     // [B][T][C] their gradients
  float* dlogits = nullptr;       // [B][T][nA]
  float* de_ws = nullptr;         // [B][h*w*C] gradient of embed_fpn.0 activations (one step)
  float* sup_final_emb = nullptr; float* sup_logits = nullptr;   // supervised step: [B][T+1][C], [B][T][nA]
  float* gpt_bwd_scratch = nullptr; size_t gpt_bwd_scratch_floats = 0;
  std::vector<jnr::ParamSeg> segs;
  std::map<std::string, int> seg_index;
  jnr::Net nets[2];                // [JN_NET_GPT_BACKBONE], [JN_NET_DETECTOR]
  bool has_net[2] = {false, false};
  int enc_net = 0;                // which net encodes patches for the GPT
  jnr::GptW gpt;
  int efpn_cin = 0, efpn_h = 0, efpn_w = 0;
  bool weights_loaded = false;
  std::vector<void*> owned;       // device allocations to free
  jnr::GptLayerPtrs* layers_dev = nullptr;
  float* emb_part = nullptr;      // [B][KS][C] split-K partials of embed_fpn.3
  int KS = 8;
  int32_t* found = nullptr;       // [B] visited bbox tiles
  float* ident = nullptr;         // identity table (scale 1, shift 0, flag 0) for gradient operands
  float* wpart = nullptr;         // [JN_NREP][JN_WPART_MAX] replicated weight-gradient partials (kept zero)
  // second stream of the conv-stack backward: the wide 1x1 weight-gradient GEMMs run beside the data-gradient GEMMs
  hipStream_t aux_stream = nullptr; hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  bool stats_prezeroed = false;   // run_net(train): the caller already zeroed the BN sums of the slots it walks
  int64_t* det_pos = nullptr; size_t det_pos_cap = 0;   // [T+1][B][2] per-step position snapshots for the detector stream
  float* det_raw = nullptr;       // [B][A][6] decoded head output
  float* det_logits = nullptr;    // [B][A][6] raw predictor outputs of the training pass (consumed by the loss kernel)
  // one entry per resident detector training pass (jn_detector_forward ... jn_detector_backward, api.hip)
  struct DetPass {
    float* dlogits = nullptr;     // [B][A][6] d loss / d raw (before the 1 / num_fg factor)
    float* acc = nullptr;         // [8] loss accumulators, [8] scale (scale[0] = loss_scale / max(num_fg, 1))
    const float* patches = nullptr; int N = 0;   // caller-owned input of the pass (the stem's weight gradient reads it)
    bool valid = false;           // cleared by every pass over the same workspace slot and by the backward
  };
  std::vector<DetPass> det_pass;
  float* det_bwd_scale = nullptr; // [1] scale of the backward in flight: forward scale x upstream d loss x chunk weight
  float* det_labels = nullptr;    // [B][nb][5] cxcywh labels of the training pass
  size_t det_labels_rows = 0;
  float* det_tmp_boxes = nullptr; int32_t* det_tmp_counts = nullptr;   // one step's detections before the scatter
  float* tok_emb = nullptr;       // [B][T][C] patch embeddings of jn_gpt_forward
  jnr::EnvState env;
  // rollout workspaces
  float* patch_emb = nullptr;     // [B, C]
  float* efpn_act = nullptr;      // [B, h*w, C]
  float* kcache = nullptr;        // [L][B][T+1][C]
  float* vcache = nullptr;
  float* logits_ws = nullptr;     // [B, nA]
  int32_t* n_done = nullptr;      // [T+1] number of finished envs after step t (index t+1)
  int64_t* prev_action = nullptr; // [B]
  int32_t* cache_len = nullptr;   // [B]
  int last_T = 0;
  bool last_stop_early = true;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool profiling = false;
  bool bwd_timed = false;         // ev[2] / ev[3] bracket a conv-stack backward
  float pdrop = 0.0f;             // --dropout (embd / attn / resid), train-mode passes only (jn_set_dropout)
  uint64_t drop_seed = 0, drop_ctr = 0, drop_seed_used = 0;   // seed of the NEXT / the most recent train-mode forward
  jn_rollout_out train_out{};     // output buffers of the most recent train-mode rollout (jn_reinforce_backward reads them)
  bool train_out_valid = false;   // cleared by every entry point that overwrites what jn_reinforce_backward reads
  // supervised autograd bridge: inputs of the most recent supervised forward (caller-owned, alive until the backward)
  struct SupState { const float* patches; const int64_t* actions; const int64_t* positions; const int64_t* classes; int B, T; } sup{};
  bool sup_valid = false;         // cleared by every pass over workspace slot 0 of the patch encoder
  jnr::ArenaSeg* segs_dev = nullptr; int segs_dev_n = 0;   // device copy of `segs` for the layout-conversion kernel
  std::vector<hipEvent_t> conv_ev;   // pairs per step when profiling
  int conv_ev_used = 0;
};
