// Host-callable launchers of the HIP kernels (internal; not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/jnroll.h"

namespace jnr {

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };
enum { JN_F32 = 0, JN_BF16 = 1 };   // storage type of an activation buffer (jn_types.h)

// Atomic accumulators (BN statistics, weight-gradient partials) are replicated JN_NREP times and a
// workgroup adds into replica (block id % JN_NREP): same-address contention drops 32x; the consumer
// (finalize / reduce kernels) sums the replicas.
constexpr int JN_NREP = 32;
constexpr int JN_WPART_MAX = 16384;   // floats per replica of the weight-gradient scratch

// Every hot-loop kernel takes (skip_flag, skip_when): it returns at once when
// *skip_flag >= skip_when.  The rollout points skip_flag at n_done[t] with skip_when = B,
// which reproduces the reference's early `break` (src/reinforce.py:181-184) with no host sync.
// Per-channel "normalize on read" table of a buffer view: a = fl ? silu(z * sc + sh) : z.
struct ChanTab {
  float* sc; float* sh; float* fl;
  // Deferred entries (train-mode forward, DESIGN.md §4 "consumer-side BatchNorm tables"): a layer whose output has at
  // most JN_DEFER_MAX_M pixels in this launch gets NO bn_finalize launch; its consumers derive (scale, shift) of such a
  // channel from the producer's batch sums in their prologue (jn_tab.h), and ONE bn_finalize_all launch at the end of the
  // pass writes the table, the saved statistics and the running averages of all those layers.  All arrays below are
  // indexed like sc / sh / fl (per view channel); dsrc == nullptr: a plain table.
  const int* dsrc;         // index of the channel's (sum, sumsq) pair inside one replica of dstats, or -1 (never deferred)
  const float* dhw;        // H * W of the layer that produced the channel (the count is dN * dhw)
  const int* dgoff; const int* dboff;   // offsets of the channel's BatchNorm weight / bias in dparams
  const float* dparams;    // parameter arena
  const double* dstats;    // [JN_NREP_DEFER used][drep_stride] batch sums of the workspace slot
  long long drep_stride;
  int dN;                  // patches in this pass
  long long dmax;          // deferral limit in pixels (dN * dhw <= dmax: deferred), jn_defer_max_m()
  // The same information as up to four channel runs in the kernel arguments (scalar registers, no memory round trip
  // before the sums can be requested): channels [c0, c1) of the view are BatchNorm channels stat0 + (c - c0) with affine
  // at g0 / b0 + (c - c0) and pixel count dN * hw — or, with stat0 < 0, plain table entries.  nseg == 0: use the arrays.
  // (four named members, not an array: an indexed array inside a by-value kernel argument sends the whole struct to
  // scratch memory — measured: 192 B / lane of scratch in every forward kernel and 30 - 50 % longer 28x28 layers)
  struct Run { int c0, c1, stat0, g0, b0; float hw; };
  int nseg;
  Run r0, r1, r2, r3;
};
constexpr long long JN_DEFER_MAX_M = 65536;   // default of jn_defer_max_m(): output pixels (N * H * W) up to which a layer's table is deferred
long long jn_defer_max_m();                   // the limit in force (env JN_DEFER_MAX_M overrides the default; read once)
constexpr int JN_NREP_DEFER = 8;              // statistics replicas such a layer accumulates into (its consumers sum them)

// Step batching of the backward: ONE launch covers `n` workspace slots (glimpse steps of a trajectory, each
// with its own batch-statistics tables); slot j adds j * stride to the per-slot pointers.  The default is the
// single-pass case.
struct SlotBatch {
  int n = 1;
  long long act = 0;      // activation elements between slots (z / x buffers)
  long long grad = 0;     // gradient floats between slots
  long long tab = 0;      // table floats between slots (the sc / sh / fl arrays each move by this)
  long long save = 0;     // saved (mean, invstd) floats between slots
  long long red = 0;      // doubles between slots of the reduction scratch
  long long consts = 0;   // floats between slots of the per-channel constants
  long long pos = 0;      // int64 elements between slots of the stem's patch positions
};

struct StemArgs {
  const float* src; const int64_t* positions; int pos_stride;   // positions[pos_stride * n + {0,1}] = (y, x)
  long long sample_stride, chan_stride; int row_stride;
  int P, N, cout;
  const float* w; void* out; int out_ld; int out_dtype;
  double* stats;                    // [JN_NREP][rep_stride]: [cout][2] sum / sumsq accumulators (train) or null
  long long stats_rep_stride;
  const int* skip_flag; int skip_when;
  int stats_nrep;                   // replicas in use (0 -> JN_NREP)
};

struct ConvArgs {
  const void* in; int in_ld; int in_dtype; ChanTab itab;
  const float* w; const float* bias;     // bias only for BN-free layers
  const void* w_bf16;                    // dense 3x3, bf16 mode: pre-rounded copy of w (or null)
  void* out; int out_ld; int out_dtype;
  int bf16_mfma;                         // 1x1 / dense 3x3: bf16 operands on v_mfma_f32_16x16x32_bf16
  int N, H, W, OH, OW, cin, cout, stride, act;
  int accumulate;                        // out += (gradient buffers)
  int w_transposed;                      // pw: w is [cin][cout] and read transposed (data-gradient)
  int in_identity;                       // the input needs no "normalize on read" (gradient views): kernels may skip the transform
  double* stats; long long stats_rep_stride;
  int stats_nrep;                        // replicas the workgroups spread their sums over (0 -> JN_NREP)
  const int* skip_flag; int skip_when;
  int n_slots;                           // pw: gridDim.z slots, pointers advance by the strides below (0 -> 1 slot)
  long long in_slot_stride, out_slot_stride, tab_slot_stride;
  const void* w_x3;                      // 1x1, fp32: the weights split into three bf16 planes (launch_w_split3), or null
  void* up_out; int up_ld;               // 1x1, fp32: also store the output nearest-x2 upsampled here (pixel stride up_ld floats), see
                                         // pw_fused_upsample_supported; null: no
};

// eval-mode DWConv (depthwise 3x3 -> BN + SiLU -> pointwise 1x1) in one kernel; mtab = table of the depthwise output
struct DwPwArgs {
  const void* in; int in_ld; ChanTab itab; const float* w_dw; ChanTab mtab; const float* w_pw;
  void* out; int out_ld; int dtype; int C, cout, N, H, W, OH, OW, stride;
  const int* skip_flag; int skip_when;
  const void* res; int res_ld; ChanTab rtab;   // optional shortcut (eval): out = silu(bn(z)) + T(res), ptab = table of z
  ChanTab ptab;
};
bool dwpw_supported(int C, int cout, int stride);
int launch_dwpw(const DwPwArgs& a, hipStream_t s);
int launch_stem(const StemArgs& a, hipStream_t s);
bool stem_rows_aligned(const StemArgs& a);        // the stem kernels may fetch the image tile with 16-byte loads
int launch_dw(const ConvArgs& a, hipStream_t s);
int launch_pw(const ConvArgs& a, hipStream_t s);
bool pw_fused_upsample_supported(const ConvArgs& a);   // launch_pw(a) with a.up_out set will take a route that writes the upsampled copy
bool pw_res_supported(const ConvArgs& a);              // kernels_pwres.hip: shapes the resident-weight 1x1 kernels take (K, N >= 64)
int launch_pw_dir(const ConvArgs& a, int ctw, int split, hipStream_t s);
int launch_pw_wide(const ConvArgs& a, hipStream_t s);   // production route: 0 when taken
bool pw_xs_supported(const ConvArgs& a);               // kernels_pwxs.hip: pixel-stationary kernel for the small maps of a forward pass
int launch_pw_xs(const ConvArgs& a, int pt, hipStream_t s, int wg_per_cu = 0);   // pt: pixel tiles per workgroup (0 = default)
int launch_pw_x3(const ConvArgs& a, int pt, hipStream_t s, int wg_per_cu = 0);   // the same on the bf16 pipe (three-way split operands); needs a.w_x3
bool pw_x3_preferred(const ConvArgs& a);
// the data gradient of a wide 1x1 layer on pw_x3_kernel (transposed split weight made on the way); -1: shape not taken
bool pw_x3_bwd_data_supported(int cout, int cin);
int launch_pw_x3_bwd_data(const ConvArgs& a, const float* w, void* w3t, hipStream_t s);
bool pw_x1_supported(const ConvArgs& a);               // bf16 inference mode: single-plane form of the same kernel
int launch_pw_x1(const ConvArgs& a, hipStream_t s);               // shapes on which the x3 kernel is the faster one
void launch_w_split3(const float* w, void* w3, long long n_floats, hipStream_t s);   // w3: 6 bytes per weight
int launch_spp(void* cat, int dtype, int ld, int h, int H, int W, int N, ChanTab it, const int* skip_flag,
               int skip_when, hipStream_t s);
int launch_upsample(const void* in, int in_ld, void* out, int out_ld, int dtype, int C, int H, int W, int N,
                    const int* skip_flag, int skip_when, hipStream_t s);
int launch_addact(const void* z, int z_ld, ChanTab zt, const void* res, int res_ld, ChanTab rt, void* out, int out_ld,
                  int dtype, int C, long long M, const int* skip_flag, int skip_when, hipStream_t s);
int launch_bn_finalize(const double* stats, long long rep_stride, double count, const float* gamma, const float* beta, float* run_mean,
                       float* run_var, float* save, ChanTab t0, ChanTab t1, int C, float eps, float momentum,
                       const int* skip_flag, int skip_when, hipStream_t s);
// deferred BatchNorm tables: one finalize launch per pass (kernels_conv.hip); arrays are per BatchNorm ("stat") channel
struct BnAllArgs {
  const double* stats; long long rep_stride; int n_stat; int N;
  const float* hw;                       // H * W of the channel's layer
  const int* goff; const int* boff;      // BatchNorm weight / bias offsets in params
  const float* params;
  const int* t0; const int* t1;          // table channel of the layer output, and of its upsampled alias (-1: none)
  float* tab; int tab_channels;          // the slot's table [3][tab_channels]
  float* save;                           // [n_stat][2] (mean, invstd)
  float* const* run_mean; float* const* run_var;   // per-channel addresses of the running statistics
  float eps, momentum;
  const int* skip_flag; int skip_when;
  long long defer_max_m;                 // layers with N * hw above it used all JN_NREP replicas
};
int launch_bn_finalize_all(const BnAllArgs& a, hipStream_t s);
int launch_nhwc_to_nchw(const void* in, int dtype, int in_ld, ChanTab it, float* out, int C, int HW, int N,
                        hipStream_t s);
int launch_efpn_linear(const float* e, const float* wt, float* part, int N, int K, int Co, int KS,
                       const int* skip_flag, int skip_when, hipStream_t s);

// ---- detector (kernels_det.hip) --------------------------------------------------------------
int launch_conv3(const ConvArgs& a, hipStream_t s);     // w_transposed: data gradient of a stride-1 layer
int launch_conv3_bwd_data_s2(const float* gz, int g_ld, const float* w, float* gin, int gin_ld, int H, int W, int OH,
                             int OW, int Co, int Ci, int N, int accumulate, hipStream_t s,
                             const SlotBatch& sb = SlotBatch{});
int launch_conv3_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw, int H,
                            int W, int OH, int OW, int Co, int Ci, int N, int stride, hipStream_t s,
                            const SlotBatch& sb = SlotBatch{});
int launch_head_pred(const void* reg, int reg_ld, ChanTab rt, const void* cls, int cls_ld, ChanTab ct, int dtype,
                     const float* wp, const float* bp, float* raw, int hid, int Hl, int Wl, int stride, int A, int a0,
                     int N, hipStream_t s, int logits_only = 0);
int launch_det_scatter(const float* boxes, const int* counts, float* out_boxes, int* out_counts, int B, int cols, int col,
                       int K, const int* skip_flag, int skip_when, hipStream_t s);
int launch_postprocess(const float* raw, int A, int N, float conf, float nms_thr, float clamp_max, float* boxes,
                       int* counts, int max_out, hipStream_t s);

// ---- detector training (kernels_detloss.hip) ------------------------------------------------------------
struct DetGeom { int A; int a0[3]; int H[3]; int W[3]; int stride[3]; };
// raw: [N][A][6] predictor outputs; labels [N][nb][5] = (class, cx, cy, w, h) floats, zero rows = padding;
// d_raw gets d loss / d raw before the 1 / max(num_fg, 1) factor, which `scale[0]` carries to the predictor backward
int launch_yolox_loss(const float* raw, const float* labels, int N, int nb, const DetGeom& geo, float* d_raw, float* acc,
                      int use_l1, float loss_scale, float* metrics, float* scale, hipStream_t s);
int launch_head_pred_bwd(const float* d_raw, const float* scale, const void* reg, int reg_ld, ChanTab rt, const void* cls,
                         int cls_ld, ChanTab ct, int dtype, const float* wp, float* g_reg, float* g_cls, float* g_wp,
                         float* g_bp, int hid, int HW, int A, int a0, int N, hipStream_t s);

// ---- backward of the conv stack (kernels_bwd.hip) ----------------------------------------
int launch_bn_bwd_reduce(const float* g, int g_ld, const void* z, int z_dtype, int z_ld, ChanTab t, const float* save,
                         int C, long long M, double* red_out, long long rep_stride, hipStream_t s,
                         const SlotBatch& sb = SlotBatch{});
// sums the replicas -> consts[c] = {sum_gy/n, sum_gy_zhat/n, gamma*invstd}; dgamma/dbeta += totals.  raw_moment: the second
// sum is sum gy * y (y = gamma * zhat + beta, formed by a consumer's fused kernel) and is converted here
int launch_bn_bwd_consts(const double* red, long long rep_stride, double count, const float* gamma, const float* beta,
                         const float* save, float* consts, float* g_gamma, float* g_beta, int C, hipStream_t s,
                         const SlotBatch& sb = SlotBatch{}, int raw_moment = 0);
int launch_bn_bwd_gz(float* g, int g_ld, const void* z, int z_dtype, int z_ld, ChanTab t, const float* save,
                     const float* consts, int C, long long M, hipStream_t s, const SlotBatch& sb = SlotBatch{});
// weight-gradient kernels add into wpart (replicated scratch, zero on entry and on exit) when the
// tensor fits, else straight into gw; launch_wpart_reduce folds the replicas into gw.
int launch_pw_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw,
                         float* wpart, long long M, int N, int K, hipStream_t s, const SlotBatch& sb = SlotBatch{},
                         long long gz_slot_stride = -1);
int launch_wpart_reduce(float* gw, float* wpart, int n, hipStream_t s);
// bn_bwd_gz + data gradient + weight gradient of a pointwise layer in one pass (few-channel layers; fp32 buffers)
struct PwBwdFusedArgs {
  const float* g; int g_ld;              // d loss / d activation of the layer output
  const float* z; int z_ld; ChanTab ot;  // raw layer output and its table
  const float* save; const float* consts;
  const float* x; int x_ld; ChanTab it;  // raw layer input and its table
  const float* w;                        // [cout][cin]
  float* gx; int gx_ld; int accumulate;  // d loss / d activation of the layer input
  float* gw; float* wpart;
  long long M; int cout, cin;
  SlotBatch sb;
  // optional: accumulate the BN-backward sums of the layer(s) that produced the input (this op writes the FINAL gradient
  // of the view): channels [0, red_split) -> red_in, [red_split, cin) -> red_in2 (either may be null: that run is
  // skipped); red_split == 0: one run over all of cin into red_in
  double* red_in; long long red_rep_stride; double* red_in2; int red_split;
  // optional: a second gradient of the layer input, added while gx is written (the shortcut branch of a bottleneck:
  // g[input] = W^T g_z + g[sum]); same pixel indexing as gx
  const float* gadd; int gadd_ld;
  // optional ("RED2"): the run [0, red_split or cin) of the input is a materialised shortcut sum silu(bn(z2)) + res — the
  // sums formed for it belong to the BatchNorm of z2's conv (the bottleneck's last pointwise conv): z2 / its table replace
  // the raw input tile and the input table in that run's silu' and second moment
  const float* red2_z; int red2_ld; const float* red2_sc; const float* red2_sh;
};
struct DwBwdFusedArgs {                  // the depthwise counterpart (3x3, pad 1, stride 1 / 2)
  const float* g; int g_ld; const float* z; int z_ld; ChanTab ot; const float* save; const float* consts;
  const float* x; int x_ld; ChanTab it; const float* w;      // w: [9][C]
  float* gin; int gin_ld; int accumulate; float* gw; float* wpart;
  int C, H, W, OH, OW, N, stride;
  SlotBatch sb;
  double* red_in; long long red_rep_stride;      // as in PwBwdFusedArgs
};
bool dw_bwd_fused_supported(int C, int H, int W, int OH, int OW, int stride);
int launch_dw_bwd_fused(const DwBwdFusedArgs& a, hipStream_t s);
bool pw_bwd_fused_supported(int cout, int cin);
bool pw_bwd_fused_reduces_input(int cout, int cin);
int launch_pw_bwd_fused(const PwBwdFusedArgs& a, hipStream_t s);
int launch_dw_bwd_data(const float* gz, int g_ld, const float* w, float* gin, int gin_ld, int C, int H, int W, int OH,
                       int OW, int N, int stride, int accumulate, hipStream_t s, const SlotBatch& sb = SlotBatch{});
int launch_dw_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw,
                         float* wpart, int C, int H, int W, int OH, int OW, int N, int stride, hipStream_t s,
                         const SlotBatch& sb = SlotBatch{});
// z != null: gz is d loss / d activation and the BN + SiLU backward (g_z) is applied while staging
int launch_stem_bwd_weight(const StemArgs& a, const float* gz, int g_ld, float* gw, float* wpart, hipStream_t s,
                           const SlotBatch& sb = SlotBatch{}, const float* z = nullptr, int z_ld = 0,
                           ChanTab ot = ChanTab{nullptr, nullptr, nullptr}, const float* save = nullptr,
                           const float* consts = nullptr);
int launch_spp_bwd(const void* cat, int dtype, float* gcat, int ld, int h, int H, int W, int N, ChanTab it,
                   hipStream_t s, const SlotBatch& sb = SlotBatch{});
int launch_upsample_bwd(const float* gdst, int dst_ld, float* gsrc, int src_ld, int C, int H, int W, int N,
                        int accumulate, hipStream_t s, const SlotBatch& sb = SlotBatch{});
int launch_grad_copy(const float* src, int src_ld, float* dst, int dst_ld, int C, long long M, int accumulate,
                     hipStream_t s, const SlotBatch& sb = SlotBatch{});
int launch_nchw_to_nhwc_grad(const float* in, float* out, int out_ld, int C, int HW, int N, int accumulate,
                             hipStream_t s);

// ---- env / rollout primitives (kernels_env.hip) -------------------------------------
// fused detection augmentation (kernels_aug.hip): params [N][AUG_NPARAM] = r_gain, b_gain, gray flag, gauss centre weight,
// gauss side weight, noise std, motion kernel [3][3] (row-major), 1 pad
#ifndef JN_DW_S2_PAD
#define JN_DW_S2_PAD 4      // LDS pixel-stride padding (floats) of the stride-2 depthwise tiles; 8 removes the bank
                            // conflicts the PMC shows (0.3-0.4 of the LDS-active cycles) but not a microsecond: 126.9 vs 127.3 ms
#endif
constexpr int AUG_NPARAM = 20;   // ... + shade intensity, shade quantity, roughness, stretch (plasma shadow), 1 pad
int launch_augment(const float* in, float* out, const float* params, const float* noise, unsigned long long seed, int N, int P,
                   hipStream_t s);
int launch_gather(const float* images, const int64_t* positions, float* out, long long out_sample_stride,
                  int B, int C, int H, int W, int P, const int* skip_flag, int skip_when, hipStream_t s,
                  const int64_t* image_index = nullptr);
int launch_bbox_masks(const int64_t* bboxes, uint8_t* masks, int32_t* n_tiles, int B, int nb, int H, int W, int P,
                      hipStream_t s);

struct EnvPtrs {
  int64_t* positions; uint8_t* bbox_masks; uint8_t* visited; int32_t* steps; uint8_t* has_stopped;
  int32_t* n_bbox_tiles; int32_t* found;
  int B, Gh, Gw, T, stop;
};
int launch_env_reset(const EnvPtrs& e, const int64_t* start_positions, uint64_t seed, hipStream_t s);
int launch_env_step(const EnvPtrs& e, const int64_t* actions, float* rewards, uint8_t* terminated,
                    uint8_t* truncated, hipStream_t s);

struct RolloutBuffers {   // device, [B,T]-shaped unless noted; any may be null
  float* rewards; float* returns; float* logprobs; float* entropies;
  uint8_t* masks;        // [B,T+1]
  uint8_t* logit_masks;
  int64_t* positions;    // [B,T+1,2]
  int64_t* actions;
  float* logits;         // [B,T,nA]
  float* final_emb;      // [B,T+1,C]
};
int launch_rollout_begin(const EnvPtrs& e, const RolloutBuffers& r, int64_t* prev_action, int32_t* cache_len,
                         int32_t* n_done, hipStream_t s);
int launch_rollout_epilogue(const RolloutBuffers& r, const int32_t* n_done, int B, int T, int stop_early,
                            hipStream_t s);

// ---- decision transformer step (kernels_gpt.hip) -------------------------------------
struct GptLayerPtrs {
  float *ln1_w, *ln1_b, *qkv_wt, *qkv_b, *proj_wt, *proj_b, *ln2_w, *ln2_b, *fc_wt, *fc_b, *fc2_wt, *fc2_b;
};

constexpr int JN_N_CLASS_ROWS = 100;   // rows of embed_class (src/models/gpt.py:227)
enum { GPT_SRC_ENV = 0, GPT_SRC_TEACH = 1, GPT_SRC_GIVEN = 2, GPT_SRC_CLASS = 3 };

struct GptStepArgs {
  int C, n_head, n_layer, nA, Tmax, B, T;
  int use_pos_emb, no_patch_emb, concat_emb, dec_pos_enc, n_parts;
  int pe2_ch;                       // channels per axis of the 2-D sinusoid table
  const float *wte, *wpe, *embed_class, *proj_wt, *proj_b, *pos1d, *pe2, *head_wt, *lnf_w, *lnf_b;
  const int64_t* classes;           // [B] class id of every agent's class token (gpt.py:476-478), or null = class 0
  const GptLayerPtrs* layers;       // device array [n_layer]
  const float* emb_part; int KS; const float* efpn_lin_b;   // patch-embedding split-K partials [B][KS][C]
  float *kcache, *vcache;           // [L][B][Tmax][C]
  int64_t* prev_action; int32_t* cache_len;
  int step;                         // t
  // token source: rollout state, teacher arrays, a given embedding row, or the class token
  int src_mode;
  int pos_index;                    // 1-D position of the token (0 in recurrent mode: gpt.py:431-449 quirk)
  const int64_t* t_actions; const int64_t* t_positions; int t_stride, t_index;
  const float* tok_emb; int tok_emb_stride, tok_emb_index;   // patch embeddings [B][stride][C]
  const float* given_emb; int given_stride, given_index;
  int embed_only;                   // stop after writing the token embedding
  int emb_stride;                   // tokens per agent in out.final_emb
  float* logits_rows; int logits_stride;   // teacher/given modes: logits of this token -> row b
  int mode; const int64_t* forced; uint64_t seed;
  EnvPtrs env;
  RolloutBuffers out;
  int32_t* n_done;                  // [T+1]
  float* tok_emb_out;               // [B][T][C] finished patch embedding of this step (training) or null
  float pdrop; uint64_t drop_seed;  // train-mode dropout (0 = off), see drop_scale in jn_device.h
  const int* skip_flag; int skip_when;
};
int launch_gpt_step(const GptStepArgs& a, hipStream_t s);
// kernels_gptmfma.hip: the same step for n_embd % 64 == 0, 4 or 16 agents per workgroup, Linears on fp32 MFMA; opt-in
// (JN_GPT_MFMA=1: measured slower, see its header); false = not taken
bool launch_gpt_step_mfma(const GptStepArgs& a, hipStream_t s);

// ---- training of the decision side (kernels_train.hip) ------------------------------------
struct LossArgs {
  const float* logits; const int64_t* actions; const float* returns; const float* rewards;
  const uint8_t* logit_masks; const int32_t* n_done;
  float* dlogits;                   // [B][T][nA]
  float* metrics;                   // [8]: action_loss, entropy_loss, loss, returns, episode_length, S
  int B, T, nA, stop_early, reward_norm;
  float ret_mean, ret_std, entropy_weight, scale;
};
int launch_reinforce_loss(const LossArgs& a, hipStream_t s);
// autograd bridge: d loss / d logits [B][T][nA] from d loss / d logprobs and d loss / d entropies [B][T] (either may be null)
int launch_logits_grad(const float* logits, const int64_t* actions, const float* dlogprobs, const float* dentropies,
                       const int32_t* n_done, float* dlogits, int B, int T, int nA, int stop_early, hipStream_t s);
// one trainable tensor of the flat arena: packing kind (api.hip PackKind) and dimensions; `off` is its offset both in the
// packed arena and in a reference-layout buffer of the same size
struct ArenaSeg { long long off, numel; int kind, d0, d1, d2; };
int launch_arena_copy(const ArenaSeg* segs, int n_segs, float* arena, float* ref, long long total, int to_ref, int accumulate,
                      hipStream_t s);

struct GptBwdArgs {
  int C, n_head, n_layer, nA, B, T, stop_early;
  int use_pos_emb, no_patch_emb, concat_emb, dec_pos_enc, pe2_ch;
  const int32_t* n_done;
  const float* final_emb;           // [B][T+1][C]
  const float* dlogits;             // [B][T][nA]
  const int64_t* actions;           // [B][T] actions taken (rollout)
  const int64_t* tok_actions;       // [B][T] action token of every patch token (teacher-forced mode) or null
  const int64_t* positions;         // [B][pos_tokens][2]
  const int64_t* classes;           // [B] class ids of the forward being differentiated, or null = class 0
  int pos_tokens;                   // T + 1 (rollout history) or T
  int pos1d_by_token;               // 1: token t has 1-D position t (full-sequence forward)
  const float* tok_emb;             // [B][T][C] patch embeddings
  float* d_tok_emb;                 // out: row of (agent b, token t) at (b * dte_stride_b + t * dte_stride_t) * C
  long long dte_stride_b, dte_stride_t;
  const float *wte, *wpe, *proj_wt, *pos1d, *pe2, *head_wt, *lnf_w, *lnf_b;
  const GptLayerPtrs* layers;       // weights
  const GptLayerPtrs* g_layers;     // gradients (same layout, non-const use)
  float *g_wte, *g_wpe, *g_embed_class, *g_proj_wt, *g_proj_b, *g_head_wt, *g_lnf_w, *g_lnf_b;
  float* scratch; long long scratch_per_agent;
  float pdrop; uint64_t drop_seed; int Tmax;   // dropout of the forward being differentiated (Tmax = block_size + 1)
};
int launch_gpt_backward(const GptBwdArgs& a, hipStream_t s);
// the same backward batched over the agents (kernels_gptbwd.hip): W / G are HOST arrays of the per-layer weight and
// gradient pointers; returns 1 when the shape is outside what it takes (caller falls back to launch_gpt_backward)
size_t gpt_backward_batched_scratch(int C, int n_head, int n_layer, int nA, int B, int T);
int launch_gpt_backward_batched(const GptBwdArgs& a, const GptLayerPtrs* W, const GptLayerPtrs* G, hipStream_t s);
int launch_ce_loss(const float* logits, const int64_t* target, const uint8_t* masks, float stop_weight, float* dlogits,
                   float* metrics, int n, int nA, int T, hipStream_t s);
// de[m][k] = 0 where e[m][k] <= 0 (ReLU mask of embed_fpn.0);  gb[o] += sum_m dpe[m][o]
int launch_relu_mask(float* de, const float* e, long long n, hipStream_t s);
int launch_colsum_add(const float* dpe, long long M, int Co, float* gb, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float wd, int step, float clip, float grad_scale, hipStream_t s);

}  // namespace jnr
