// C ABI of libjnroll.so: context, weight packing, env, networks and the rollout loop.
// Host code only (compiled by hipcc as C++); kernels live in kernels_*.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <memory>
#include <set>

#include "jn_internal.h"

namespace jnr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int build_pafpn(Net& net, std::vector<ParamEntry>& params, const std::string& prefix, float depth, float width,
                bool depthwise, int P);
int build_head(Net& net, std::vector<ParamEntry>& params, const std::string& prefix, float width, bool depthwise,
               int num_classes);

namespace {

void add_param(std::vector<ParamEntry>& params, const std::string& name, std::initializer_list<int64_t> shape,
               int dtype, bool buffer, bool used) {
  ParamEntry e;
  std::memset(&e, 0, sizeof(e));
  std::snprintf(e.info.name, sizeof(e.info.name), "%s", name.c_str());
  e.info.dtype = dtype;
  e.info.ndim = (int)shape.size();
  int i = 0;
  for (auto s : shape) e.info.shape[i++] = s;
  e.info.is_buffer = buffer;
  e.info.used = used;
  params.push_back(e);
}

template <typename T>
int dev_alloc(jn_ctx* ctx, T** out, size_t count) {
  void* p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(&p, count * sizeof(T));
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    return JN_ENOMEM;
  }
  ctx->owned.push_back(p);
  *out = (T*)p;
  return JN_OK;
}

int dev_upload(jn_ctx* ctx, float** out, const std::vector<float>& host) {
  if (!*out) {                                   // re-uploads (weights changed) reuse the allocation
    int rc = dev_alloc(ctx, out, host.size());
    if (rc) return rc;
  }
  JN_HIP(hipMemcpy(*out, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return JN_OK;
}

// ---- flat parameter store ---------------------------------------------------------------
// Every trainable tensor lives in one contiguous fp32 arena (packed layouts), mirrored by a
// gradient arena and the AdamW moments: the optimiser and the RCCL all-reduce see ONE buffer.
enum PackKind { PK_RAW = 0, PK_T, PK_STEM, PK_DW, PK_CONV3, PK_EFPN_LIN };

int store_param(jn_ctx* ctx, const std::string& name, const std::vector<float>& packed, int kind, int d0, int d1,
                int d2, float** out) {
  auto it = ctx->seg_index.find(name);
  if (it == ctx->seg_index.end()) {
    const size_t padded = (packed.size() + 3) / 4 * 4;
    // tensors of 64 values and more start on a multiple of 8 floats (pw_x3_kernel's split weights come in groups of 8; the
    // small ones — predictor rows and biases — stay back to back, the head kernels read them as one matrix)
    if (packed.size() >= 64) ctx->arena_used = (ctx->arena_used + 7) / 8 * 8;
    JN_CHECK(ctx->params && ctx->arena_used + padded <= ctx->arena_size, JN_ENOMEM, "parameter arena exhausted at '%s'",
             name.c_str());
    ParamSeg sg;
    sg.name = name; sg.kind = kind; sg.d0 = d0; sg.d1 = d1; sg.d2 = d2; sg.off = ctx->arena_used; sg.numel = packed.size();
    ctx->arena_used += padded;
    ctx->seg_index[name] = (int)ctx->segs.size();
    ctx->segs.push_back(sg);
    it = ctx->seg_index.find(name);
  }
  const ParamSeg& sg = ctx->segs[it->second];
  JN_CHECK(sg.numel == packed.size(), JN_EINVAL, "'%s' changed size between loads", name.c_str());
  JN_HIP(hipMemcpy(ctx->params + sg.off, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = ctx->params + sg.off;
  return JN_OK;
}

// inverse of the packing: arena layout -> the reference's (PyTorch) layout
std::vector<float> unpack_param(const ParamSeg& sg, const std::vector<float>& p) {
  std::vector<float> t(p.size());
  switch (sg.kind) {
    case PK_T:        // stored [in][out] -> [out][in]; d0 = out, d1 = in
      for (int o = 0; o < sg.d0; ++o)
        for (int i = 0; i < sg.d1; ++i) t[(size_t)o * sg.d1 + i] = p[(size_t)i * sg.d0 + o];
      break;
    case PK_STEM:     // [(c*6+dy)*6+dx][oc] -> [oc][q*3+c][ky][kx]; d0 = cout
      for (int oc = 0; oc < sg.d0; ++oc)
        for (int c = 0; c < 3; ++c)
          for (int dy = 0; dy < 6; ++dy)
            for (int dx = 0; dx < 6; ++dx) {
              const int ky = dy >> 1, py = dy & 1, kx = dx >> 1, px = dx & 1, q = py + 2 * px;
              t[(((size_t)oc * 12 + q * 3 + c) * 3 + ky) * 3 + kx] = p[(size_t)((c * 6 + dy) * 6 + dx) * sg.d0 + oc];
            }
      break;
    case PK_DW:       // [tap][c] -> [c][tap]; d0 = C
      for (int c = 0; c < sg.d0; ++c)
        for (int k = 0; k < 9; ++k) t[(size_t)c * 9 + k] = p[(size_t)k * sg.d0 + c];
      break;
    case PK_CONV3:    // [tap][o][k] -> [o][k][tap]; d0 = cout, d1 = cin
      for (int o = 0; o < sg.d0; ++o)
        for (int k = 0; k < sg.d1; ++k)
          for (int tp = 0; tp < 9; ++tp) t[((size_t)o * sg.d1 + k) * 9 + tp] = p[((size_t)tp * sg.d0 + o) * sg.d1 + k];
      break;
    case PK_EFPN_LIN: // [(p*C + ch)][o] -> [o][ch*HW + p]; d0 = C(out), d1 = HW, d2 = C(in)
      for (int o = 0; o < sg.d0; ++o)
        for (int ch = 0; ch < sg.d2; ++ch)
          for (int q = 0; q < sg.d1; ++q)
            t[(size_t)o * sg.d1 * sg.d2 + (size_t)ch * sg.d1 + q] = p[((size_t)q * sg.d2 + ch) * sg.d0 + o];
      break;
    default: t = p;
  }
  return t;
}

struct TensorMap {
  std::map<std::string, const jn_tensor*> m;
  const float* f32(const std::string& name, size_t numel) const {
    auto it = m.find(name);
    if (it == m.end()) { set_error("state-dict entry '%s' is missing", name.c_str()); return nullptr; }
    const jn_tensor* t = it->second;
    size_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= (size_t)t->shape[i];
    if (t->dtype != 0 || n != numel) {
      set_error("state-dict entry '%s' has %zu elements / dtype %d, expected %zu float32", name.c_str(), n, t->dtype, numel);
      return nullptr;
    }
    return (const float*)t->data;
  }
};

// Linear weight [out][in] -> transposed [in][out]
std::vector<float> transpose(const float* w, int out, int in) {
  std::vector<float> t((size_t)out * in);
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) t[(size_t)i * out + o] = w[(size_t)o * in + i];
  return t;
}

int upload_raw(jn_ctx* ctx, const TensorMap& tm, const std::string& name, size_t n, float** out) {
  const float* p = tm.f32(name, n);
  if (!p) return JN_ENOTFOUND;
  return store_param(ctx, name, std::vector<float>(p, p + n), PK_RAW, (int)n, 0, 0, out);
}
int upload_buf(jn_ctx* ctx, const TensorMap& tm, const std::string& name, size_t n, float** out) {
  const float* p = tm.f32(name, n);
  if (!p) return JN_ENOTFOUND;
  return dev_upload(ctx, out, std::vector<float>(p, p + n));
}
int upload_t(jn_ctx* ctx, const TensorMap& tm, const std::string& name, int out_f, int in_f, float** out) {
  const float* p = tm.f32(name, (size_t)out_f * in_f);
  if (!p) return JN_ENOTFOUND;
  return store_param(ctx, name, transpose(p, out_f, in_f), PK_T, out_f, in_f, 0, out);
}

// YOLOX BaseConv = bias-free conv + BatchNorm2d(eps=1e-3, momentum=0.03) + SiLU.
// Nothing is folded: convs write raw z and consumers apply (scale, shift) + SiLU on read.
constexpr float kBnEps = 1e-3f;
constexpr float kBnMomentum = 0.03f;

// The two halves of a merged pair (ConvW::prefix2): every tensor kind is stored first-half then second-half, back to
// back, so the kernels see one conv of cout channels.
int pack_conv_pair(jn_ctx* ctx, const TensorMap& tm, ConvW& cw) {
  const int h = cw.cout_first, h2 = cw.cout - h;
  int rc;
  auto pair_raw = [&](const std::string& leaf, size_t n1, size_t n2, float** out) -> int {
    float *a = nullptr, *b = nullptr;
    if ((rc = upload_raw(ctx, tm, cw.prefix + leaf, n1, &a))) return rc;
    if ((rc = upload_raw(ctx, tm, cw.prefix2 + leaf, n2, &b))) return rc;
    JN_CHECK(b == a + n1, JN_ESTATE, "merged conv pair '%s': halves of %s are not contiguous in the arena", cw.prefix.c_str(), leaf.c_str());
    *out = a;
    return JN_OK;
  };
  auto pair_buf = [&](const std::string& leaf, float** out) -> int {
    const float* a = tm.f32(cw.prefix + leaf, h);
    const float* b = tm.f32(cw.prefix2 + leaf, h2);
    if (!a || !b) return JN_ENOTFOUND;
    std::vector<float> both(a, a + h);
    both.insert(both.end(), b, b + h2);
    return dev_upload(ctx, out, both);
  };
  if ((rc = pair_raw(".bn.weight", h, h2, &cw.gamma_dev))) return rc;
  if ((rc = pair_raw(".bn.bias", h, h2, &cw.beta_dev))) return rc;
  if ((rc = pair_buf(".bn.running_mean", &cw.rmean_dev))) return rc;
  if ((rc = pair_buf(".bn.running_var", &cw.rvar_dev))) return rc;
  return pair_raw(".conv.weight", (size_t)h * cw.cin, (size_t)h2 * cw.cin, &cw.w_dev);
}

int pack_conv(jn_ctx* ctx, const TensorMap& tm, ConvW& cw, OpKind kind) {
  if (!cw.prefix2.empty()) return pack_conv_pair(ctx, tm, cw);
  const int cig = cw.cin / cw.groups;
  const std::string wname = cw.prefix + (cw.has_bn ? ".conv.weight" : ".weight");
  const float* w = tm.f32(wname, (size_t)cw.cout * cig * cw.k * cw.k);
  if (!w) return JN_ENOTFOUND;
  int rc;
  if (cw.has_bn) {
    const int co = cw.cout;
    if ((rc = upload_raw(ctx, tm, cw.prefix + ".bn.weight", co, &cw.gamma_dev))) return rc;
    if ((rc = upload_raw(ctx, tm, cw.prefix + ".bn.bias", co, &cw.beta_dev))) return rc;
    if ((rc = upload_buf(ctx, tm, cw.prefix + ".bn.running_mean", co, &cw.rmean_dev))) return rc;
    if ((rc = upload_buf(ctx, tm, cw.prefix + ".bn.running_var", co, &cw.rvar_dev))) return rc;
  } else if (cw.has_bias) {
    if ((rc = upload_raw(ctx, tm, cw.prefix + ".bias", cw.cout, &cw.b_dev))) return rc;
  }
  std::vector<float> packed;
  int pk = PK_RAW;
  if (kind == OP_STEM) {
    pk = PK_STEM;
    // [oc][q*3 + c][ky][kx] (Focus order TL, BL, TR, BR: q = py + 2*px)  ->  [(c*6+dy)*6+dx][oc]
    packed.assign((size_t)108 * cw.cout, 0.0f);
    for (int oc = 0; oc < cw.cout; ++oc)
      for (int c = 0; c < 3; ++c)
        for (int dy = 0; dy < 6; ++dy)
          for (int dx = 0; dx < 6; ++dx) {
            const int ky = dy >> 1, py = dy & 1, kx = dx >> 1, px = dx & 1, q = py + 2 * px;
            packed[(size_t)((c * 6 + dy) * 6 + dx) * cw.cout + oc] = w[(((size_t)oc * 12 + q * 3 + c) * 3 + ky) * 3 + kx];
          }
  } else if (kind == OP_DW) {
    pk = PK_DW;
    packed.resize((size_t)9 * cw.cout);
    for (int c = 0; c < cw.cout; ++c)
      for (int t = 0; t < 9; ++t) packed[(size_t)t * cw.cout + c] = w[(size_t)c * 9 + t];
  } else if (kind == OP_PW) {
    packed.assign(w, w + (size_t)cw.cout * cw.cin);
  } else if (kind == OP_CONV3) {
    // [tap][oc][cin]: every tap is a 1x1 GEMM weight
    pk = PK_CONV3;
    packed.resize((size_t)9 * cw.cout * cw.cin);
    for (int o = 0; o < cw.cout; ++o)
      for (int k = 0; k < cw.cin; ++k)
        for (int t = 0; t < 9; ++t) packed[((size_t)t * cw.cout + o) * cw.cin + k] = w[((size_t)o * cw.cin + k) * 9 + t];
  } else {
    return JN_OK;
  }
  int rcs = store_param(ctx, wname, packed, pk, cw.cout, cw.cin, 0, &cw.w_dev);
  if (rcs == JN_OK && kind == OP_CONV3 && ctx->cfg.act_dtype == JN_BF16) {
    // bf16 inference mode: the MFMA operands are bf16 anyway — round the weights once here (round to nearest even)
    // instead of in every workgroup's staging loop
    std::vector<uint16_t> hb(packed.size());
    for (size_t i = 0; i < packed.size(); ++i) {
      uint32_t u;
      std::memcpy(&u, &packed[i], 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      hb[i] = (uint16_t)(u >> 16);
    }
    if (!cw.w_bf16) {
      uint16_t* d = nullptr;
      if ((rcs = dev_alloc(ctx, &d, hb.size()))) return rcs;
      cw.w_bf16 = d;
    }
    JN_HIP(hipMemcpy(cw.w_bf16, hb.data(), hb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  return rcs;
}

// get_emb(pos * inv_freq) of positional_encodings >= 6 (interleaved sin, cos), SURVEY.md §2.2
std::vector<float> sinusoid_row(int pos, int channels) {
  std::vector<float> r(channels);
  for (int i = 0; i < channels; i += 2) {
    const float inv_freq = 1.0f / std::pow(10000.0f, (float)i / (float)channels);
    const float ang = (float)pos * inv_freq;
    r[i] = std::sin(ang);
    if (i + 1 < channels) r[i + 1] = std::cos(ang);
  }
  return r;
}

int n_parts(const jn_config& c) { return 2 + (c.no_patch_emb ? 0 : 1) + (c.use_pos_emb ? 1 : 0); }

}  // namespace

// ---- workspace slots ----------------------------------------------------------------
// Activation buffers are stored as net.act_dtype (JN_F32: 4-byte, JN_BF16: 2-byte elements).
inline size_t act_esz(const Net& net) { return net.act_dtype == JN_BF16 ? 2 : 4; }
inline void* view_ptr(const Net& net, int slot, int max_batch, const View& v) {
  const size_t elem = (size_t)slot * net.per_image_floats * max_batch + net.buf_off[v.buf] * (size_t)max_batch + v.coff;
  return reinterpret_cast<char*>(net.act) + elem * act_esz(net);
}
inline ChanTab view_tab(const Net& net, int slot, const View& v) {
  float* t = net.tab + (size_t)slot * 3 * net.tab_channels + net.tab_off[v.buf] + v.coff;
  return ChanTab{t, t + net.tab_channels, t + 2 * net.tab_channels};
}
inline double* slot_stats(const Net& net, int slot) { return net.stats + (size_t)slot * JN_NREP * 2 * net.stat_channels; }
inline float* slot_save(const Net& net, int slot) { return net.save + (size_t)slot * 2 * net.stat_channels; }

}  // namespace jnr

using namespace jnr;


extern "C" {

int jn_abi_version(void) { return JN_ABI_VERSION; }
const char* jn_last_error(void) { return jnr::g_err; }

int jn_create(const jn_config* cfg, jn_ctx** out) {
  JN_CHECK(cfg && out, JN_EINVAL, "jn_create: null argument");
  JN_CHECK(cfg->struct_size == (int)sizeof(jn_config), JN_EINVAL, "jn_config.struct_size %d != %zu", cfg->struct_size,
           sizeof(jn_config));
  JN_CHECK(cfg->n_embd > 0 && cfg->n_embd % 4 == 0 && cfg->n_embd <= 256, JN_EINVAL,
           "n_embd %d unsupported (multiple of 4, <= 256)", cfg->n_embd);
  JN_CHECK(cfg->n_head > 0 && cfg->n_embd % cfg->n_head == 0, JN_EINVAL, "n_embd %% n_head != 0");
  JN_CHECK(cfg->n_actions == 8 || cfg->n_actions == 9, JN_EINVAL, "n_actions must be 8 or 9");
  JN_CHECK(cfg->block_size >= 1 && cfg->block_size <= 255, JN_EINVAL, "block_size out of range");
  JN_CHECK(cfg->max_batch >= 1, JN_EINVAL, "max_batch must be >= 1");
  JN_CHECK(cfg->gpt_bb_width > 0 || cfg->with_detector || cfg->no_patch_emb, JN_EINVAL,
           "no patch encoder: set gpt_bb_width or with_detector (or no_patch_emb)");
  std::unique_ptr<jn_ctx> ctx(new jn_ctx());
  ctx->cfg = *cfg;
  JN_CHECK(cfg->act_dtype == JN_F32 || cfg->act_dtype == JN_BF16, JN_EINVAL, "act_dtype must be 0 (fp32) or 1 (bf16)");
  ctx->nets[0].act_dtype = ctx->nets[1].act_dtype = cfg->act_dtype;
  if (ctx->cfg.det_nms_threshold <= 0) ctx->cfg.det_nms_threshold = 0.45f;
  if (ctx->cfg.max_det_per_patch <= 0) ctx->cfg.max_det_per_patch = 64;
  const int C = cfg->n_embd, nA = cfg->n_actions;
  auto& P = ctx->params_tab;
  // ---- state-dict table in the reference's construction order (src/models/gpt.py:221-318) ----
  add_param(P, "action_head.lm_heads.0.weight", {nA, C}, 0, false, true);
  {
    const int ch2 = (int)std::ceil(C / 4.0) * 2;
    add_param(P, "positional_encoding.inv_freq", {(ch2 + 1) / 2}, 0, true, false);
    if (cfg->decoder_pos_encoding) {
      const int ch1 = (int)std::ceil(C / 2.0) * 2;
      add_param(P, "decoder_token_pos_enc.inv_freq", {ch1 / 2}, 0, true, false);
    }
  }
  add_param(P, "embed_class.weight", {100, C}, 0, false, true);
  if (cfg->concat_emb) {
    add_param(P, "project_concat.weight", {C, (int64_t)n_parts(*cfg) * C}, 0, false, true);
    add_param(P, "project_concat.bias", {C}, 0, false, true);
  }
  int rc;
  if (cfg->with_detector) {
    rc = build_pafpn(ctx->nets[JN_NET_DETECTOR], P, "yolox.backbone.", cfg->det_depth, cfg->det_width,
                     cfg->det_depthwise != 0, cfg->patch_size);
    if (rc) return rc;
    ctx->has_net[JN_NET_DETECTOR] = true;
    rc = build_head(ctx->nets[JN_NET_DETECTOR], P, "yolox.head.", cfg->det_width, cfg->det_depthwise != 0, 1);
    if (rc) return rc;
  }
  if (cfg->gpt_bb_width > 0) {
    rc = build_pafpn(ctx->nets[JN_NET_GPT_BACKBONE], P, "gpt_backbone.", cfg->gpt_bb_depth, cfg->gpt_bb_width,
                     cfg->gpt_bb_depthwise != 0, cfg->patch_size);
    if (rc) return rc;
    ctx->has_net[JN_NET_GPT_BACKBONE] = true;
    ctx->enc_net = JN_NET_GPT_BACKBONE;
  } else {
    ctx->enc_net = JN_NET_DETECTOR;
  }
  if (!cfg->no_patch_emb) {
    const Net& enc = ctx->nets[ctx->enc_net];
    ctx->efpn_cin = enc.fpn[2].C; ctx->efpn_h = enc.fpn[2].H; ctx->efpn_w = enc.fpn[2].W;
    // split-K slices of embed_fpn.3: ~192 inputs each, 8..64 slices (their partials are summed by the consumer)
    ctx->KS = std::max(8, std::min(64, (ctx->efpn_h * ctx->efpn_w * C + 191) / 192));
    add_param(P, "embed_fpn.0.weight", {C, ctx->efpn_cin, 1, 1}, 0, false, true);
    add_param(P, "embed_fpn.3.weight", {C, (int64_t)ctx->efpn_h * ctx->efpn_w * C}, 0, false, true);
    add_param(P, "embed_fpn.3.bias", {C}, 0, false, true);
  }
  add_param(P, "transformer.wte.weight", {nA, C}, 0, false, true);
  add_param(P, "transformer.wpe.weight", {cfg->pos_emb_size > 0 ? cfg->pos_emb_size : 1, C}, 0, false,
            !cfg->decoder_pos_encoding);
  const int bs1 = cfg->block_size + 1;
  for (int l = 0; l < cfg->n_layer; ++l) {
    const std::string p = "transformer.h." + std::to_string(l) + ".";
    add_param(P, p + "ln_1.weight", {C}, 0, false, true);
    add_param(P, p + "ln_1.bias", {C}, 0, false, true);
    add_param(P, p + "attn.c_attn.weight", {3 * C, C}, 0, false, true);
    add_param(P, p + "attn.c_attn.bias", {3 * C}, 0, false, true);
    add_param(P, p + "attn.c_proj.weight", {C, C}, 0, false, true);
    add_param(P, p + "attn.c_proj.bias", {C}, 0, false, true);
    add_param(P, p + "attn.bias", {1, 1, bs1, bs1}, 0, true, false);
    add_param(P, p + "ln_2.weight", {C}, 0, false, true);
    add_param(P, p + "ln_2.bias", {C}, 0, false, true);
    add_param(P, p + "mlp.c_fc.weight", {4 * C, C}, 0, false, true);
    add_param(P, p + "mlp.c_fc.bias", {4 * C}, 0, false, true);
    add_param(P, p + "mlp.c_proj.weight", {C, 4 * C}, 0, false, true);
    add_param(P, p + "mlp.c_proj.bias", {C}, 0, false, true);
  }
  add_param(P, "transformer.ln_f.weight", {C}, 0, false, true);
  add_param(P, "transformer.ln_f.bias", {C}, 0, false, true);
  *out = ctx.release();
  return JN_OK;
}

int jn_destroy(jn_ctx* ctx) {
  if (!ctx) return JN_OK;
  (void)hipSetDevice(ctx->cfg.device);
  (void)hipDeviceSynchronize();
  for (void* p : ctx->owned) (void)hipFree(p);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->conv_ev) (void)hipEventDestroy(e);
  if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  delete ctx;
  return JN_OK;
}

int jn_param_count(const jn_ctx* ctx) { return ctx ? (int)ctx->params_tab.size() : JN_EINVAL; }

int jn_param_info_at(const jn_ctx* ctx, int index, jn_param_info* out) {
  JN_CHECK(ctx && out && index >= 0 && index < (int)ctx->params_tab.size(), JN_EINVAL, "jn_param_info_at: bad index %d", index);
  *out = ctx->params_tab[index].info;
  return JN_OK;
}

// (scale, shift, flag) of BN channels from the running statistics (eval mode)
__global__ void bn_eval_table_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rmean, const float* __restrict__ rvar, ChanTab t0,
                                     ChanTab t1, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rvar[c] + eps);
  const float sh = beta[c] - rmean[c] * sc;
  t0.sc[c] = sc; t0.sh[c] = sh; t0.fl[c] = 1.0f;
  if (t1.sc) { t1.sc[c] = sc; t1.sh[c] = sh; t1.fl[c] = 1.0f; }
}

__global__ void fill_kernel(float* p, float v, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// Allocates (or grows to) n_slots workspace slots of a net; tables start as identity
// (scale 1, shift 0, flag 0 = "already an activation").
static int ensure_slots(jn_ctx* ctx, Net& net, int n_slots) {
  if (net.n_slots >= n_slots) return JN_OK;
  const int MB = ctx->cfg.max_batch;
  float *tab = nullptr, *save = nullptr;
  double* stats = nullptr;
  char* act = nullptr;
  int rc;
  if ((rc = dev_alloc(ctx, &act, (size_t)n_slots * net.per_image_floats * MB * act_esz(net)))) return rc;
  if ((rc = dev_alloc(ctx, &tab, (size_t)n_slots * 3 * net.tab_channels))) return rc;
  if ((rc = dev_alloc(ctx, &save, (size_t)n_slots * 2 * net.stat_channels))) return rc;
  if ((rc = dev_alloc(ctx, &stats, (size_t)n_slots * JN_NREP * 2 * net.stat_channels))) return rc;
  // old (smaller) allocations stay owned by the context until jn_destroy; slots are grown once per config.  What the
  // old slots hold moves along (a detector pass may need its slot AFTER a train-mode rollout filled the encoder's, and
  // the rollout's backward still reads them: the reference's statement order, src/reinforce.py:326-341)
  const int n_old = net.n_slots;
  if (n_old > 0) {
    JN_HIP(hipDeviceSynchronize());
    JN_HIP(hipMemcpy(act, net.act, (size_t)n_old * net.per_image_floats * MB * act_esz(net), hipMemcpyDeviceToDevice));
    JN_HIP(hipMemcpy(tab, net.tab, (size_t)n_old * 3 * net.tab_channels * sizeof(float), hipMemcpyDeviceToDevice));
    JN_HIP(hipMemcpy(save, net.save, (size_t)n_old * 2 * net.stat_channels * sizeof(float), hipMemcpyDeviceToDevice));
    JN_HIP(hipMemcpy(stats, net.stats, (size_t)n_old * JN_NREP * 2 * net.stat_channels * sizeof(double), hipMemcpyDeviceToDevice));
  }
  net.act = act; net.tab = tab; net.save = save; net.stats = stats; net.n_slots = n_slots;
  for (int sl = n_old; sl < n_slots; ++sl) {
    float* t = tab + (size_t)sl * 3 * net.tab_channels;
    const long long n = net.tab_channels;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, t, 1.0f, n);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, 0, t + n, 0.0f, 2 * n);
  }
  JN_HIP(hipGetLastError());
  JN_HIP(hipDeviceSynchronize());           // (the fills ran on the null stream; callers launch on theirs)
  if (n_old == 0) net.eval_tab_dirty = true;
  return JN_OK;
}

// Slot-0 table from the BN running statistics (after a weight load or a training step).
static int refresh_eval_table(jn_ctx* ctx, Net& net, hipStream_t s) {
  if (!net.eval_tab_dirty) return JN_OK;
  for (const Op& op : net.ops) {
    if (op.wslot < 0) continue;
    const ConvW& cw = net.convs[op.wslot];
    if (!cw.has_bn) continue;
    ChanTab t1{nullptr, nullptr, nullptr};
    if (op.alias.buf >= 0) t1 = view_tab(net, 0, op.alias);
    hipLaunchKernelGGL(bn_eval_table_kernel, dim3((cw.cout + 63) / 64), dim3(64), 0, s, cw.gamma_dev, cw.beta_dev,
                       cw.rmean_dev, cw.rvar_dev, view_tab(net, 0, op.out), t1, cw.cout, kBnEps);
  }
  JN_HIP(hipGetLastError());
  net.eval_tab_dirty = false;
  return JN_OK;
}

// Device-side workspaces; allocated on the first jn_load_weights (needs a GPU).
static int alloc_workspaces(jn_ctx* ctx) {
  jn_ctx& x = *ctx;
  const jn_config& c = ctx->cfg;
  const int B = c.max_batch, C = c.n_embd;
  int rc;
  for (int n = 0; n < 2; ++n) {
    if (!ctx->has_net[n]) continue;
    if ((rc = ensure_slots(ctx, ctx->nets[n], 1))) return rc;
  }
  const int Tmax = c.block_size + 1;
  if ((rc = dev_alloc(ctx, &ctx->kcache, (size_t)c.n_layer * B * Tmax * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->vcache, (size_t)c.n_layer * B * Tmax * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->n_done, (size_t)Tmax + 1))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->prev_action, (size_t)B))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->cache_len, (size_t)B))) return rc;
  if ((rc = dev_alloc(ctx, &x.emb_part, (size_t)B * x.KS * C))) return rc;
  if (!c.no_patch_emb)
    if ((rc = dev_alloc(ctx, &ctx->efpn_act, (size_t)B * ctx->efpn_h * ctx->efpn_w * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->patch_emb, (size_t)B * C))) return rc;
  JN_HIP(hipMemset(ctx->n_done, 0, ((size_t)Tmax + 1) * sizeof(int32_t)));
  for (auto& e : ctx->ev) JN_HIP(hipEventCreate(&e));
  return JN_OK;
}

int jn_load_weights(jn_ctx* ctx, const jn_tensor* tensors, size_t n) {
  JN_CHECK(ctx && tensors, JN_EINVAL, "jn_load_weights: null argument");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  TensorMap tm;
  for (size_t i = 0; i < n; ++i) tm.m[tensors[i].name] = &tensors[i];
  int rc;
  if (!ctx->kcache) {
    if ((rc = alloc_workspaces(ctx))) return rc;
  }
  if (!ctx->params) {
    size_t total = 0;
    for (const ParamEntry& e : ctx->params_table()) {
      if (e.info.dtype != 0 || e.info.is_buffer || !e.info.used) continue;
      size_t n = 1;
      for (int i = 0; i < e.info.ndim; ++i) n *= (size_t)e.info.shape[i];
      total += (n + 3) / 4 * 4 + (n >= 64 ? 4 : 0);
    }
    ctx->arena_size = total;
    if ((rc = dev_alloc(ctx, &ctx->params, total))) return rc;
    JN_HIP(hipMemset(ctx->params, 0, total * sizeof(float)));
    if ((rc = dev_alloc(ctx, &ctx->params_x3, 3 * total))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->params_x3t, 3 * total))) return rc;
  }
  const jn_config& c = ctx->cfg;
  const int C = c.n_embd, nA = c.n_actions;
  auto pack_net = [&](int ni) -> int {
    if (!ctx->has_net[ni]) return JN_OK;
    Net& net = ctx->nets[ni];
    for (const Op& op : net.ops) {
      if (op.wslot < 0) continue;
      int r = pack_conv(ctx, tm, net.convs[op.wslot], op.kind);
      if (r) return r;
    }
    net.eval_tab_dirty = true;
    net.x3_dirty = true;
    for (const Op& op : net.ops) {
      if (op.kind != OP_PRED) continue;
      const std::string k = std::to_string(op.level), hp = "yolox.head.";
      const int hid = net.head_hid;
      const float* rw = tm.f32(hp + "reg_preds." + k + ".weight", (size_t)4 * hid);
      const float* rb = tm.f32(hp + "reg_preds." + k + ".bias", 4);
      const float* ow = tm.f32(hp + "obj_preds." + k + ".weight", hid);
      const float* ob = tm.f32(hp + "obj_preds." + k + ".bias", 1);
      const float* cw = tm.f32(hp + "cls_preds." + k + ".weight", hid);
      const float* cb = tm.f32(hp + "cls_preds." + k + ".bias", 1);
      if (!rw || !rb || !ow || !ob || !cw || !cb) return JN_ENOTFOUND;
      // arena-resident (trainable): reg (4 x hid) | obj (hid) | cls (hid) rows back to back = one [6][hid] matrix;
      // biases reg (4) | obj (1, padded to 4) | cls (1, padded to 4): entries 0..3, 4 and 8 of `pred_b`
      (void)rw; (void)rb; (void)ow; (void)ob; (void)cw; (void)cb;
      int r;
      float *w_reg = nullptr, *w_obj = nullptr, *w_cls = nullptr, *b_reg = nullptr, *b_obj = nullptr, *b_cls = nullptr;
      if ((r = upload_raw(ctx, tm, hp + "reg_preds." + k + ".weight", (size_t)4 * hid, &w_reg))) return r;
      if ((r = upload_raw(ctx, tm, hp + "obj_preds." + k + ".weight", hid, &w_obj))) return r;
      if ((r = upload_raw(ctx, tm, hp + "cls_preds." + k + ".weight", hid, &w_cls))) return r;
      if ((r = upload_raw(ctx, tm, hp + "reg_preds." + k + ".bias", 4, &b_reg))) return r;
      if ((r = upload_raw(ctx, tm, hp + "obj_preds." + k + ".bias", 1, &b_obj))) return r;
      if ((r = upload_raw(ctx, tm, hp + "cls_preds." + k + ".bias", 1, &b_cls))) return r;
      JN_CHECK(w_obj == w_reg + 4 * hid && w_cls == w_obj + hid && b_obj == b_reg + 4 && b_cls == b_reg + 8, JN_ESTATE,
               "predictor tensors of level %d are not contiguous in the arena", op.level);
      net.pred_w[op.level] = w_reg;
      net.pred_b[op.level] = b_reg;
    }
    return JN_OK;
  };
  // arena order: [gpt_backbone | decision model] = what optim_gpt updates (gpt.py:552-557), then yolox.*
  if ((rc = pack_net(JN_NET_GPT_BACKBONE))) return rc;
  GptW& g = ctx->gpt;
  if ((rc = upload_raw(ctx, tm, "transformer.wte.weight", (size_t)nA * C, &g.wte))) return rc;
  if (!c.decoder_pos_encoding) {
    if ((rc = upload_raw(ctx, tm, "transformer.wpe.weight", (size_t)std::max(c.pos_emb_size, 1) * C, &g.wpe))) return rc;
  }
  if ((rc = upload_raw(ctx, tm, "embed_class.weight", (size_t)100 * C, &g.embed_class))) return rc;
  if (c.concat_emb) {
    if ((rc = upload_t(ctx, tm, "project_concat.weight", C, n_parts(c) * C, &g.proj_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, "project_concat.bias", C, &g.proj_b))) return rc;
  }
  {
    const int ch1 = (int)std::ceil(C / 2.0) * 2;
    const int Tmax = c.block_size + 1;
    std::vector<float> p1((size_t)Tmax * C);
    for (int t = 0; t < Tmax; ++t) {
      std::vector<float> r = sinusoid_row(t, ch1);
      std::copy(r.begin(), r.begin() + C, p1.begin() + (size_t)t * C);
    }
    if ((rc = dev_upload(ctx, &g.pos1d, p1))) return rc;
    const int ch2 = (int)std::ceil(C / 4.0) * 2;
    std::vector<float> tab((size_t)256 * ch2);
    for (int p = 0; p < 256; ++p) {
      std::vector<float> r = sinusoid_row(p, ch2);
      std::copy(r.begin(), r.end(), tab.begin() + (size_t)p * ch2);
    }
    if ((rc = dev_upload(ctx, &g.pos2d_col, tab))) return rc;
  }
  if (!c.no_patch_emb) {
    if ((rc = upload_raw(ctx, tm, "embed_fpn.0.weight", (size_t)C * ctx->efpn_cin, &g.efpn_w))) return rc;
    const int HW = ctx->efpn_h * ctx->efpn_w;
    const float* lw = tm.f32("embed_fpn.3.weight", (size_t)C * HW * C);
    if (!lw) return JN_ENOTFOUND;
    // Flatten order of the reference is (c, h, w) (nn.Flatten on NCHW, gpt.py:304); ours is (h, w, c).
    std::vector<float> wt((size_t)HW * C * C);
    for (int o = 0; o < C; ++o)
      for (int ch = 0; ch < C; ++ch)
        for (int p = 0; p < HW; ++p) wt[((size_t)p * C + ch) * C + o] = lw[(size_t)o * HW * C + (size_t)ch * HW + p];
    if ((rc = store_param(ctx, "embed_fpn.3.weight", wt, PK_EFPN_LIN, C, HW, C, &g.efpn_lin_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, "embed_fpn.3.bias", C, &g.efpn_lin_b))) return rc;
  }
  if ((rc = upload_t(ctx, tm, "action_head.lm_heads.0.weight", nA, C, &g.head_wt))) return rc;
  if ((rc = upload_raw(ctx, tm, "transformer.ln_f.weight", C, &g.lnf_w))) return rc;
  if ((rc = upload_raw(ctx, tm, "transformer.ln_f.bias", C, &g.lnf_b))) return rc;
  g.layers.resize(c.n_layer);
  std::vector<GptLayerPtrs> lp(c.n_layer);
  for (int l = 0; l < c.n_layer; ++l) {
    const std::string p = "transformer.h." + std::to_string(l) + ".";
    GptW::Layer& L = g.layers[l];
    if ((rc = upload_raw(ctx, tm, p + "ln_1.weight", C, &L.ln1_w))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "ln_1.bias", C, &L.ln1_b))) return rc;
    if ((rc = upload_t(ctx, tm, p + "attn.c_attn.weight", 3 * C, C, &L.qkv_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "attn.c_attn.bias", 3 * C, &L.qkv_b))) return rc;
    if ((rc = upload_t(ctx, tm, p + "attn.c_proj.weight", C, C, &L.proj_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "attn.c_proj.bias", C, &L.proj_b))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "ln_2.weight", C, &L.ln2_w))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "ln_2.bias", C, &L.ln2_b))) return rc;
    if ((rc = upload_t(ctx, tm, p + "mlp.c_fc.weight", 4 * C, C, &L.fc_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "mlp.c_fc.bias", 4 * C, &L.fc_b))) return rc;
    if ((rc = upload_t(ctx, tm, p + "mlp.c_proj.weight", C, 4 * C, &L.fc2_wt))) return rc;
    if ((rc = upload_raw(ctx, tm, p + "mlp.c_proj.bias", C, &L.fc2_b))) return rc;
    lp[l] = GptLayerPtrs{L.ln1_w, L.ln1_b, L.qkv_wt, L.qkv_b, L.proj_wt, L.proj_b,
                         L.ln2_w, L.ln2_b, L.fc_wt, L.fc_b, L.fc2_wt, L.fc2_b};
  }
  {
    GptLayerPtrs* d = nullptr;
    if ((rc = dev_alloc(ctx, &d, (size_t)c.n_layer))) return rc;
    JN_HIP(hipMemcpy(d, lp.data(), lp.size() * sizeof(GptLayerPtrs), hipMemcpyHostToDevice));
    ctx->layers_dev = d;
  }
  if (!ctx->gpt_arena_end) ctx->gpt_arena_end = ctx->arena_used;
  if ((rc = pack_net(JN_NET_DETECTOR))) return rc;
  ctx->det_head_begin = ctx->arena_used;
  for (const ParamSeg& sg : ctx->segs)
    if (sg.name.compare(0, 11, "yolox.head.") == 0) { ctx->det_head_begin = std::min(ctx->det_head_begin, sg.off); }
  JN_HIP(hipDeviceSynchronize());
  ctx->weights_loaded = true;
  return JN_OK;
}

// ---- network execution ---------------------------------------------------------------
static View net_full_view(const Net& net, int buf) {
  View v; v.buf = buf; v.H = net.bufs[buf].H; v.W = net.bufs[buf].W; v.C = net.bufs[buf].C; v.coff = 0;
  return v;
}

struct StemSrc {
  const float* src; const int64_t* positions; long long sample_stride, chan_stride; int row_stride;
  int pos_stride = 2;
};
static inline int det_slot_base(const jn_ctx* ctx);   // first workspace slot of the detector's training passes (below)


// Descriptors of the deferred BatchNorm tables (ChanTab): which (sum, sumsq) pair, BatchNorm weight / bias and pixel
// count stand behind every table channel, and the reverse map for the one finalize launch per pass.  Only the depthwise
// fp32 PAFPN (the nano patch encoder) takes part: its consumers all read their table through jn_tab.h.
static int ensure_defer_tables(jn_ctx* ctx, Net& net) {
  if (net.defer_built) return JN_OK;
  net.defer_built = true;
  static const bool off = std::getenv("JN_NO_DEFER_BN") != nullptr;
  bool ok = !off && net.depthwise && net.act_dtype == JN_F32 && ctx->params;
  const int n_ops = net.n_backbone_ops < 0 ? (int)net.ops.size() : net.n_backbone_ops;
  for (int oi = 0; ok && oi < n_ops; ++oi) {
    const Op& op = net.ops[oi];
    if (op.kind == OP_CONV3 || op.kind == OP_PRED) ok = false;
    if (op.kind == OP_DW && op.in.C % 16 != 0) ok = false;
    if (op.wslot >= 0 && !net.convs[op.wslot].has_bn) ok = false;
  }
  if (!ok) return JN_OK;
  const int TC = net.tab_channels, SC = net.stat_channels;
  std::vector<int> td_src(TC, -1), td_g(TC, 0), td_b(TC, 0), fd_g(SC, 0), fd_b(SC, 0), fd_t0(SC, 0), fd_t1(SC, -1);
  std::vector<float> td_hw(TC, 0.0f), fd_hw(SC, 0.0f);
  std::vector<float*> fd_rm(SC, nullptr), fd_rv(SC, nullptr);
  for (int oi = 0; oi < n_ops; ++oi) {
    const Op& op = net.ops[oi];
    if (op.wslot < 0) continue;
    const ConvW& cw = net.convs[op.wslot];
    const float hw = (float)(op.out.H * op.out.W);
    for (int j = 0; j < cw.cout; ++j) {
      const int i = cw.stat_off + j;
      const int g = (int)(cw.gamma_dev - ctx->params) + j, b = (int)(cw.beta_dev - ctx->params) + j;
      fd_hw[i] = hw; fd_g[i] = g; fd_b[i] = b; fd_rm[i] = cw.rmean_dev + j; fd_rv[i] = cw.rvar_dev + j;
      const View* vs[2] = {&op.out, op.alias.buf >= 0 ? &op.alias : nullptr};
      for (int k = 0; k < 2; ++k) {
        if (!vs[k]) continue;
        const int tc = net.tab_off[vs[k]->buf] + vs[k]->coff + j;
        (k == 0 ? fd_t0 : fd_t1)[i] = tc;
        td_src[tc] = i; td_hw[tc] = hw; td_g[tc] = g; td_b[tc] = b;
      }
    }
  }
  int rc;
  auto up = [&](auto** dst, const auto& host) -> int {
    using T = typename std::remove_reference<decltype(host[0])>::type;
    typename std::remove_const<T>::type* d = nullptr;
    if ((rc = dev_alloc(ctx, &d, host.size()))) return rc;
    JN_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = d;
    return JN_OK;
  };
  if ((rc = up(&net.td_src, td_src)) || (rc = up(&net.td_goff, td_g)) || (rc = up(&net.td_boff, td_b)) || (rc = up(&net.td_hw, td_hw)) ||
      (rc = up(&net.fd_hw, fd_hw)) || (rc = up(&net.fd_goff, fd_g)) || (rc = up(&net.fd_boff, fd_b)) || (rc = up(&net.fd_t0, fd_t0)) ||
      (rc = up(&net.fd_t1, fd_t1)) || (rc = up(&net.fd_rm, fd_rm)) || (rc = up(&net.fd_rv, fd_rv)))
    return rc;
  net.h_td_hw.assign(TC, 0.0f);
  for (int tc = 0; tc < TC; ++tc) if (td_src[tc] >= 0) net.h_td_hw[tc] = td_hw[tc];
  net.h_td_src = td_src; net.h_td_goff = td_g; net.h_td_boff = td_b;
  net.defer_ok = true;
  return JN_OK;
}

// One pass of a PAFPN over N patches in workspace slot `slot`.  train != 0: batch-statistics
// BatchNorm (stats accumulated by every conv, finalised per layer, running stats updated).
static int run_net(jn_ctx* ctx, int ni, int N, const StemSrc& ss, int slot, int train, const int* skip_flag,
                   int skip_when, hipStream_t s, bool with_head = false, int first_op = 0) {
  Net& net = ctx->nets[ni];
  const int n_ops = (with_head || net.n_backbone_ops < 0) ? (int)net.ops.size() : net.n_backbone_ops;
  const int MB = ctx->cfg.max_batch;
  int rc;
  // the autograd bridges differentiate LATER what a forward left in the workspace: a pass over the same slots in
  // between makes that state stale (the backward entry points then fail with JN_ESTATE instead of computing garbage)
  // (slots: 0 = eval / supervised pass; 1 .. T = the glimpse steps of a train-mode rollout of the ENCODER net; from
  // det_slot_base on = detector training passes)
  if (ni == ctx->enc_net) {
    if (slot == 0) ctx->sup_valid = false;
    else if (ni != JN_NET_DETECTOR || slot < det_slot_base(ctx)) ctx->train_out_valid = false;
  }
  if (ni == JN_NET_DETECTOR && slot >= det_slot_base(ctx) && slot - det_slot_base(ctx) < (int)ctx->det_pass.size())
    ctx->det_pass[slot - det_slot_base(ctx)].valid = false;
  if (!train && (rc = refresh_eval_table(ctx, net, s))) return rc;
  // fp32 passes: the 1x1 weights of this net as three bf16 planes for pw_x3_kernel (bf16 inference mode uses the h plane
  // alone: pw_x1).  Split again only after something wrote the arena (jn_load_weights, jn_import_arena, an optimiser
  // step: mark_params_written) — a rollout's T passes and every eval pass in between reuse the planes (ADVICE round 3:
  // for yolox-s / -m detectors the split is tens of MB per pass)
  const bool x3 = ctx->params_x3 && !std::getenv("JN_NO_PW_X3");   // read per pass: tests flip it
  if (x3) {
    if (net.x3_hi == 0) {
      size_t lo = ctx->arena_size, hi = 0;
      for (const Op& op : net.ops) {
        if (op.kind != OP_PW || op.wslot < 0) continue;
        const ConvW& cw = net.convs[op.wslot];
        if (!cw.w_dev) continue;
        const size_t o = (size_t)(cw.w_dev - ctx->params);
        lo = std::min(lo, o); hi = std::max(hi, o + (size_t)cw.cout * cw.cin);
      }
      net.x3_lo = lo / 8 * 8; net.x3_hi = hi > lo ? (hi + 7) / 8 * 8 : 0;
    }
    if (net.x3_hi > net.x3_lo && net.x3_dirty) {
      launch_w_split3(ctx->params + net.x3_lo, ctx->params_x3 + 3 * net.x3_lo, (long long)(net.x3_hi - net.x3_lo), s);
      net.x3_dirty = false;
    }
  }
  double* stats = train ? slot_stats(net, slot) : nullptr;
  float* save = train ? slot_save(net, slot) : nullptr;
  // (a train-mode rollout zeroes the statistics of all its slots with ONE memset up front)
  if (train && !ctx->stats_prezeroed) JN_HIP(hipMemsetAsync(stats, 0, (size_t)JN_NREP * 2 * net.stat_channels * sizeof(double), s));
  const long long rep_stride = 2LL * net.stat_channels;
  // deferred tables: the small-map layers of the depthwise fp32 encoder get no finalize launch of their own; their
  // consumers read the batch sums (ChanTab in jn_kernels.h), one finalize launch closes the pass
  if (train && !with_head && (rc = ensure_defer_tables(ctx, net))) return rc;
  const bool defer = train && !with_head && net.defer_ok;
  auto deferred = [&](const Op& op) { return defer && (long long)N * op.out.H * op.out.W <= jn_defer_max_m(); };
  auto ptr = [&](const View& v) { return view_ptr(net, slot, MB, v); };
  auto tab = [&](const View& v) {
    ChanTab t = view_tab(net, slot, v);
    bool any = false;                       // does the view hold a channel whose table is deferred in this pass?
    if (defer) {
      const int off = net.tab_off[v.buf] + v.coff;
      for (int c = 0; c < v.C && !any; ++c) any = net.h_td_hw[off + c] > 0.0f && (double)N * net.h_td_hw[off + c] <= (double)jn_defer_max_m();
    }
    if (any) {
      const int off = net.tab_off[v.buf] + v.coff;
      t.dsrc = net.td_src + off; t.dhw = net.td_hw + off; t.dgoff = net.td_goff + off; t.dboff = net.td_boff + off;
      t.dparams = ctx->params; t.dstats = stats; t.drep_stride = rep_stride; t.dN = N; t.dmax = jn_defer_max_m();
      // channel runs for the kernel arguments (at most four: a concat of a few producers); more: the arrays above
      int ns = 0;
      bool fits = true;
      for (int c = 0; c < v.C && fits;) {
        const int tc = off + c;
        const bool d = net.h_td_hw[tc] > 0.0f && (double)N * net.h_td_hw[tc] <= (double)jn_defer_max_m();
        int e = c + 1;
        if (d) {
          while (e < v.C && net.h_td_src[off + e] == net.h_td_src[tc] + (e - c) && net.h_td_goff[off + e] == net.h_td_goff[tc] + (e - c) &&
                 net.h_td_boff[off + e] == net.h_td_boff[tc] + (e - c) && net.h_td_hw[off + e] == net.h_td_hw[tc])
            ++e;
        } else {
          while (e < v.C && !(net.h_td_hw[off + e] > 0.0f && (double)N * net.h_td_hw[off + e] <= (double)jn_defer_max_m())) ++e;
        }
        if (ns == 4) { fits = false; break; }
        const ChanTab::Run run{c, e, d ? net.h_td_src[tc] : -1, net.h_td_goff[tc], net.h_td_boff[tc], net.h_td_hw[tc]};
        (ns == 0 ? t.r0 : ns == 1 ? t.r1 : ns == 2 ? t.r2 : t.r3) = run;
        ++ns;
        c = e;
      }
      t.nseg = fits ? ns : 0;
    }
    return t;
  };
  auto ld = [&](const View& v) { return net.bufs[v.buf].C; };
  auto finalize = [&](const Op& op, const ConvW& cw) {
    if (!train || !cw.has_bn || deferred(op)) return;
    ChanTab t1{nullptr, nullptr, nullptr};
    if (op.alias.buf >= 0) t1 = tab(op.alias);
    // with the end-of-pass finalize (defer): table only here, saved / running statistics there
    launch_bn_finalize(stats + 2 * cw.stat_off, rep_stride, (double)N * op.out.H * op.out.W, cw.gamma_dev, cw.beta_dev,
                       defer ? nullptr : cw.rmean_dev, defer ? nullptr : cw.rvar_dev, defer ? nullptr : save + 2 * cw.stat_off,
                       tab(op.out), t1, cw.cout, kBnEps, kBnMomentum, skip_flag, skip_when, s);
  };
  // JN_LAYER_PROFILE=1: HIP events around every op of the pass, table on stderr (a measuring aid, off by default)
  static const bool layer_profile = std::getenv("JN_LAYER_PROFILE") != nullptr;
  std::vector<hipEvent_t> lev;
  if (layer_profile) {
    lev.resize(n_ops + 1);
    for (auto& e : lev) (void)hipEventCreate(&e);
    (void)hipEventRecord(lev[first_op], s);
  }
  std::set<int> fused_ups;                  // upsample ops whose copy the producing 1x1 kernel wrote
  for (int oi = first_op; oi < n_ops; ++oi) {
    const Op& op = net.ops[oi];
    switch (op.kind) {
      case OP_STEM: {
        const ConvW& cw = net.convs[op.wslot];
        StemArgs a{ss.src, ss.positions, ss.pos_stride, ss.sample_stride, ss.chan_stride, ss.row_stride, net.P, N, cw.cout,
                   cw.w_dev, ptr(op.out), ld(op.out), net.act_dtype, train ? stats + 2 * cw.stat_off : nullptr, rep_stride,
                   skip_flag, skip_when, deferred(op) ? JN_NREP_DEFER : JN_NREP};
        launch_stem(a, s);
        finalize(op, cw);
        break;
      }
      case OP_PW:
      case OP_CONV3:
      case OP_DW: {
        const ConvW& cw = net.convs[op.wslot];
        static const bool no_fused_eval = std::getenv("JN_NO_FUSED_EVAL") != nullptr;
        if (!train && !no_fused_eval && net.act_dtype == JN_F32 && op.kind == OP_DW && oi + 1 < n_ops) {
          // eval: DWConv = depthwise + pointwise in one kernel, the depthwise output stays on chip
          const Op& nx = net.ops[oi + 1];
          if (nx.kind == OP_PW && nx.in.buf == op.out.buf && nx.in.coff == op.out.coff && nx.in.C == op.out.C &&
              dwpw_supported(op.out.C, nx.out.C, op.stride)) {
            const ConvW& pw = net.convs[nx.wslot];
            DwPwArgs f{};
            f.in = ptr(op.in); f.in_ld = ld(op.in); f.itab = tab(op.in); f.w_dw = cw.w_dev; f.mtab = tab(op.out);
            f.w_pw = pw.w_dev; f.out = ptr(nx.out); f.out_ld = ld(nx.out); f.dtype = net.act_dtype;
            f.C = op.out.C; f.cout = nx.out.C; f.N = N; f.H = op.in.H; f.W = op.in.W; f.OH = op.out.H; f.OW = op.out.W;
            f.stride = op.stride; f.skip_flag = skip_flag; f.skip_when = skip_when;
            int skip_ops = 1;
            if (oi + 2 < n_ops && net.ops[oi + 2].kind == OP_ADDACT && net.ops[oi + 2].in.buf == nx.out.buf &&
                net.ops[oi + 2].in.coff == nx.out.coff) {
              // bottleneck shortcut: the add + activation goes into the epilogue, the pconv's raw z is never stored
              const Op& ad = net.ops[oi + 2];
              f.res = ptr(ad.res); f.res_ld = ld(ad.res); f.rtab = tab(ad.res); f.ptab = tab(nx.out);
              f.out = ptr(ad.out); f.out_ld = ld(ad.out);
              skip_ops = 2;
            }
            launch_dwpw(f, s);
            if (layer_profile) for (int k = 1; k <= skip_ops + 1; ++k) (void)hipEventRecord(lev[oi + k], s);
            oi += skip_ops;
            continue;
          }
        }
        ConvArgs a{};
        a.in = ptr(op.in); a.in_ld = ld(op.in); a.in_dtype = net.act_dtype; a.itab = tab(op.in); a.w = cw.w_dev;
        a.w_bf16 = cw.w_bf16;
        if (x3 && op.kind == OP_PW && (cw.w_dev - ctx->params) % 8 == 0) a.w_x3 = ctx->params_x3 + 3 * (cw.w_dev - ctx->params);
        a.bias = cw.b_dev; a.out = ptr(op.out); a.out_ld = ld(op.out); a.out_dtype = net.act_dtype;
        a.bf16_mfma = net.act_dtype == JN_BF16;
        a.N = N; a.H = op.in.H; a.W = op.in.W; a.OH = op.out.H; a.OW = op.out.W;
        a.cin = op.in.C; a.cout = op.out.C; a.stride = op.stride; a.act = op.act;
        a.stats = (train && cw.has_bn) ? stats + 2 * cw.stat_off : nullptr;
        a.stats_rep_stride = rep_stride;
        a.stats_nrep = deferred(op) ? JN_NREP_DEFER : JN_NREP;
        a.skip_flag = skip_flag; a.skip_when = skip_when;
        if (op.kind == OP_PW && net.act_dtype == JN_F32) {
          // the source of a nearest x2 upsample: the producing kernel writes the upsampled copy itself where its route can
          for (int uj = oi + 1; uj < n_ops; ++uj) {
            const Op& up = net.ops[uj];
            if (up.kind == OP_UPSAMPLE && up.in.buf == op.out.buf && up.in.coff == op.out.coff && up.in.C == op.out.C) {
              if (pw_fused_upsample_supported(a)) { a.up_out = ptr(up.out); a.up_ld = ld(up.out); fused_ups.insert(uj); }
              break;
            }
          }
        }
        if (op.kind == OP_PW) { JN_CHECK(launch_pw(a, s) == 0, JN_ESTATE, "1x1 conv %s: no kernel for this shape", op.name.c_str()); }
        else if (op.kind == OP_DW) launch_dw(a, s); else launch_conv3(a, s);
        finalize(op, cw);
        break;
      }
      case OP_SPP:
        launch_spp(view_ptr(net, slot, MB, net_full_view(net, op.out.buf)), net.act_dtype, ld(op.out), op.in.C, op.in.H, op.in.W, N,
                   tab(op.in), skip_flag, skip_when, s);
        break;
      case OP_UPSAMPLE:
        if (!fused_ups.count(oi))
          launch_upsample(ptr(op.in), ld(op.in), ptr(op.out), ld(op.out), net.act_dtype, op.in.C, op.in.H, op.in.W, N, skip_flag,
                          skip_when, s);
        break;
      case OP_ADDACT:
        launch_addact(ptr(op.in), ld(op.in), tab(op.in), ptr(op.res), ld(op.res), tab(op.res), ptr(op.out), ld(op.out),
                      net.act_dtype, op.out.C, (long long)N * op.out.H * op.out.W, skip_flag, skip_when, s);
        break;
      case OP_PRED:
        launch_head_pred(ptr(op.in), ld(op.in), tab(op.in), ptr(op.res), ld(op.res), tab(op.res), net.act_dtype, net.pred_w[op.level],
                         net.pred_b[op.level], train ? ctx->det_logits : ctx->det_raw, net.head_hid, op.in.H, op.in.W, op.stride,
                         net.n_anchors, op.anchor0, N, s, train ? 1 : 0);
        break;
    }
    if (layer_profile) (void)hipEventRecord(lev[oi + 1], s);
  }
  if (layer_profile) {
    (void)hipStreamSynchronize(s);
    static const char* kn[] = {"stem", "pw", "dw", "conv3", "spp", "upsample", "addact", "pred"};
    const double esz = (double)act_esz(net);
    double tot_us = 0, tot_b = 0;
    fprintf(stderr, "# layer profile: net %d, N=%d, train=%d, slot=%d\n", ni, N, train, slot);
    for (int oi = first_op; oi < n_ops; ++oi) {
      const Op& op = net.ops[oi];
      float ms = 0;
      (void)hipEventElapsedTime(&ms, lev[oi], lev[oi + 1]);
      const double in_e = op.kind == OP_STEM ? 3.0 * net.P * net.P * 4.0 / esz : (double)op.in.H * op.in.W * op.in.C;
      double out_e = (double)op.out.H * op.out.W * op.out.C;
      double elems = in_e + out_e;
      if (op.kind == OP_ADDACT) elems += out_e;
      if (op.kind == OP_SPP) elems = in_e * 4;
      const double bytes = elems * esz * N;
      tot_us += ms * 1e3; tot_b += bytes;
      fprintf(stderr, "%-8s %-44s in %3dx%3dx%3d out %3dx%3dx%3d s%d  %8.1f us  %7.1f MB  %7.0f GB/s\n", kn[op.kind], op.name.c_str(),
              op.in.H, op.in.W, op.in.C, op.out.H, op.out.W, op.out.C, op.stride, ms * 1e3, bytes / 1e6, bytes / (ms * 1e-3) / 1e9);
    }
    fprintf(stderr, "# total %.1f us, %.1f MB, %.0f GB/s\n", tot_us, tot_b / 1e6, tot_b / (tot_us * 1e-6) / 1e9);
    for (auto& e : lev) (void)hipEventDestroy(e);
  }
  if (defer) {
    BnAllArgs fa{};
    fa.stats = stats; fa.rep_stride = rep_stride; fa.n_stat = net.stat_channels; fa.N = N; fa.hw = net.fd_hw;
    fa.goff = net.fd_goff; fa.boff = net.fd_boff; fa.params = ctx->params; fa.t0 = net.fd_t0; fa.t1 = net.fd_t1;
    fa.tab = net.tab + (size_t)slot * 3 * net.tab_channels; fa.tab_channels = net.tab_channels; fa.save = save;
    fa.run_mean = net.fd_rm; fa.run_var = net.fd_rv; fa.eps = kBnEps; fa.momentum = kBnMomentum;
    fa.skip_flag = skip_flag; fa.skip_when = skip_when; fa.defer_max_m = jn_defer_max_m();
    launch_bn_finalize_all(fa, s);
  }
  JN_HIP(hipGetLastError());
  if (train) net.eval_tab_dirty = true;     // running statistics moved
  return JN_OK;
}

// ---- training state --------------------------------------------------------------------
static int ensure_train_state(jn_ctx* ctx, int g_slots = 1) {
  int rc;
  if (!ctx->grads) {
    if ((rc = dev_alloc(ctx, &ctx->grads, ctx->arena_size))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->adam_m, ctx->arena_size))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->adam_v, ctx->arena_size))) return rc;
    JN_HIP(hipMemset(ctx->grads, 0, ctx->arena_size * sizeof(float)));
    JN_HIP(hipMemset(ctx->adam_m, 0, ctx->arena_size * sizeof(float)));
    JN_HIP(hipMemset(ctx->adam_v, 0, ctx->arena_size * sizeof(float)));
    std::vector<float> id(3 * 2048, 0.0f);
    for (int i = 0; i < 2048; ++i) id[i] = 1.0f;
    if ((rc = dev_upload(ctx, &ctx->ident, id))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->wpart, (size_t)JN_NREP * JN_WPART_MAX))) return rc;
    JN_HIP(hipMemset(ctx->wpart, 0, (size_t)JN_NREP * JN_WPART_MAX * sizeof(float)));
  }
  for (int ni = 0; ni < 2; ++ni) {
    if (!ctx->has_net[ni]) continue;
    Net& net = ctx->nets[ni];
    const int want = ni == ctx->enc_net ? g_slots : 1;
    if (net.g_slots >= want) continue;
    // (a smaller earlier allocation stays owned by the context until jn_destroy)
    if ((rc = dev_alloc(ctx, &net.gact, (size_t)want * net.per_image_floats * ctx->cfg.max_batch))) return rc;
    if ((rc = dev_alloc(ctx, &net.bred, (size_t)want * JN_NREP * 2 * net.stat_channels))) return rc;
    if ((rc = dev_alloc(ctx, &net.bconsts, (size_t)want * 3 * net.stat_channels))) return rc;
    net.g_slots = want;
  }
  return JN_OK;
}

static inline float* grad_of(const jn_ctx* ctx, const float* param) { return ctx->grads + (param - ctx->params); }

// The context's second stream (non-blocking) with its fork / join events: independent kernel families run beside the
// caller's stream — weight-gradient GEMMs in the backward, the detector beside the next glimpse step in a rollout.
static int ensure_aux_stream(jn_ctx* ctx) {
  if (ctx->aux_stream) return JN_OK;
  JN_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  JN_HIP(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
  JN_HIP(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
  return JN_OK;
}

static inline bool views_overlap(const View& a, const View& b) {
  return a.buf >= 0 && a.buf == b.buf && a.coff < b.coff + b.C && b.coff < a.coff + a.C;
}

// Backward fusion of a producer's BN reduce into its consumer: returns the index of the ONE BatchNorm conv whose
// output view is exactly ops[obi].in, provided ops[obi] is the only reader of that view, nothing else writes into it
// and no gradient arrives from outside the network (FPN outputs); -1 otherwise.
// fpn_zero: bit i set = NO gradient arrives in net.fpn[i] from outside the network in this backward (the REINFORCE /
// supervised backward only feeds fpn[2]); such a view is an ordinary single-consumer output.
static int sole_producer(const Net& net, int obi, int fpn_zero = 0) {
  const View& v = net.ops[obi].in;
  for (int i = 0; i < 3; ++i)
    if (!((fpn_zero >> i) & 1) && views_overlap(v, net.fpn[i])) return -1;
  int prod = -1;
  for (int j = 0; j < (int)net.ops.size(); ++j) {
    const Op& o = net.ops[j];
    if (j != obi && (views_overlap(o.in, v) || views_overlap(o.res, v))) return -1;
    if (o.kind == OP_SPP && o.out.buf == v.buf) return -1;            // works on the whole concat buffer
    if (views_overlap(o.out, v) || views_overlap(o.alias, v)) {
      if (prod >= 0 || j >= obi) return -1;
      prod = j;
    }
  }
  if (prod < 0) return -1;
  const Op& po = net.ops[prod];
  if (po.wslot < 0 || po.out.coff != v.coff || po.out.C != v.C || po.out.buf != v.buf) return -1;
  const ConvW& pw = net.convs[po.wslot];
  if (!pw.has_bn || !pw.prefix2.empty() || pw.cout != v.C) return -1;
  return prod;
}

// Generalisation for the fused 1x1 kernels (round 3): the producer of channel segment [coff, coff + C) of buffer `buf` as
// ONE BatchNorm conv — its whole output, or one half of a merged conv2|conv1 pair (half 0 = conv2 rows, 1 = conv1 rows) —
// whose only reader is op `reader` (plus, optionally, the shortcut add `folded_addact` whose gradient copy the reader's
// kernel has absorbed).  The reader then holds the FINAL gradient of the segment and can form that layer's BN-backward
// sums in its epilogue.
struct RedRun { int wslot = -1; int half = -1; };     // half -1: whole conv
static bool red_segment(const Net& net, int reader, int buf, int coff, int C, int folded_addact, RedRun& out, int fpn_zero = 0) {
  View seg; seg.buf = buf; seg.coff = coff; seg.C = C;
  for (int i = 0; i < 3; ++i)
    if (!((fpn_zero >> i) & 1) && views_overlap(seg, net.fpn[i])) return false;
  int prod = -1;
  for (int j = 0; j < (int)net.ops.size(); ++j) {
    const Op& o = net.ops[j];
    if (j != reader && views_overlap(o.in, seg)) return false;
    if (j != reader && j != folded_addact && views_overlap(o.res, seg)) return false;
    if (o.kind == OP_SPP && o.out.buf == buf) return false;
    if (views_overlap(o.out, seg) || views_overlap(o.alias, seg)) {
      if (prod >= 0 || j >= reader) return false;
      prod = j;
    }
  }
  if (prod < 0) return false;
  const Op& po = net.ops[prod];
  if (po.wslot < 0 || po.out.buf != buf || views_overlap(po.alias, seg)) return false;
  const ConvW& pw = net.convs[po.wslot];
  if (!pw.has_bn) return false;
  if (pw.prefix2.empty()) {
    if (po.out.coff != coff || po.out.C != C || pw.cout != C) return false;
    out.wslot = po.wslot; out.half = -1;
    return true;
  }
  if (pw.cout != 2 * C || 2 * pw.cout_first != pw.cout || po.out.C != 2 * C) return false;
  if (coff == po.out.coff) out.half = 0; else if (coff == po.out.coff + C) out.half = 1; else return false;
  out.wslot = po.wslot;
  return true;
}

// RED2 (round 3): segment [coff, coff + C) of `buf` is a materialised shortcut sum (the output of an OP_ADDACT) read only by
// op `reader` (and, as residual, by the shortcut add `folded_addact` whose gradient copy the reader's kernel absorbs): the
// reader writes the FINAL gradient of the sum, which is also the gradient of the activation of the add's `in` operand —
// the bottleneck's last pointwise conv.  Returns that conv's op index (BatchNorm, whole output == the add's `in`, no other
// reader) and the add's index, or -1.
static int shortcut_sum_conv(const Net& net, int reader, int buf, int coff, int C, int folded_addact, int* addact_out) {
  View seg; seg.buf = buf; seg.coff = coff; seg.C = C;
  for (int i = 0; i < 3; ++i)
    if (views_overlap(seg, net.fpn[i])) return -1;
  int add = -1;
  for (int j = 0; j < (int)net.ops.size(); ++j) {
    const Op& o = net.ops[j];
    if (j != reader && views_overlap(o.in, seg)) return -1;
    if (j != reader && j != folded_addact && views_overlap(o.res, seg)) return -1;
    if (o.kind == OP_SPP && o.out.buf == buf) return -1;
    if (views_overlap(o.out, seg) || views_overlap(o.alias, seg)) {
      if (add >= 0 || j >= reader) return -1;
      add = j;
    }
  }
  if (add < 0 || net.ops[add].kind != OP_ADDACT) return -1;
  const Op& ao = net.ops[add];
  if (ao.out.coff != coff || ao.out.C != C || ao.in.C != C) return -1;
  int conv = -1;
  for (int j = 0; j < (int)net.ops.size(); ++j) {
    const Op& o = net.ops[j];
    if (j != add && (views_overlap(o.in, ao.in) || views_overlap(o.res, ao.in))) return -1;
    if (j == add && views_overlap(o.res, ao.in)) return -1;
    if (views_overlap(o.out, ao.in) || views_overlap(o.alias, ao.in)) {
      if (conv >= 0 || j >= add) return -1;
      conv = j;
    }
  }
  if (conv < 0) return -1;
  const Op& co = net.ops[conv];
  if (co.wslot < 0 || co.kind != OP_PW || views_overlap(co.alias, ao.in)) return -1;
  const ConvW& cw = net.convs[co.wslot];
  if (!cw.has_bn || !cw.prefix2.empty() || co.out.buf != ao.in.buf || co.out.coff != ao.in.coff || co.out.C != C || cw.cout != C) return -1;
  *addact_out = add;
  return conv;
}

// Backward of `nsl` train-mode PAFPN passes (workspace slots slot .. slot + nsl - 1, N patches each; gradient
// slots 0 .. nsl - 1): every kernel is launched ONCE for all the passes (SlotBatch), so a 20-step trajectory
// costs the launches of one pass.  g[fpn views] must hold the incoming gradients; parameter gradients are
// accumulated into ctx->grads.  ss.positions belongs to the first pass, pos_slot_stride int64s separate passes.
static int run_net_backward(jn_ctx* ctx, int ni, int N, const StemSrc& ss, int slot, hipStream_t s, int nsl = 1,
                            long long pos_slot_stride = 0, bool with_head = false, int fpn_zero = 0) {
  Net& net = ctx->nets[ni];
  const int MB = ctx->cfg.max_batch;
  JN_CHECK(nsl >= 1 && nsl <= net.g_slots && slot + nsl <= net.n_slots, JN_ESTATE, "backward over %d slots from %d: not allocated", nsl, slot);
  JN_HIP(hipMemsetAsync(net.bred, 0, (size_t)nsl * JN_NREP * 2 * net.stat_channels * sizeof(double), s));
  const long long rep_stride = 2LL * net.stat_channels;
  SlotBatch sb;
  sb.n = nsl;
  sb.act = (long long)net.per_image_floats * MB;
  sb.grad = (long long)net.per_image_floats * MB;
  sb.tab = 3LL * net.tab_channels;
  sb.save = 2LL * net.stat_channels;
  sb.red = (long long)JN_NREP * 2 * net.stat_channels;
  sb.consts = 3LL * net.stat_channels;
  sb.pos = pos_slot_stride;
  float* save = slot_save(net, slot);
  auto ptr = [&](const View& v) { return view_ptr(net, slot, MB, v); };
  auto gptr = [&](const View& v) { return net.gact + net.buf_off[v.buf] * (size_t)MB + v.coff; };
  auto tab = [&](const View& v) { return view_tab(net, slot, v); };
  auto ld = [&](const View& v) { return net.bufs[v.buf].C; };
  const ChanTab ident{ctx->ident, ctx->ident + 2048, ctx->ident + 4096};
  std::map<int, std::pair<int, View>> g_alias;      // conv output buffer -> (coff, gradient view to read instead)
  std::set<int> red_done;                           // conv slots whose BN-backward sums a consumer's kernel already formed
  std::map<int, int> red_half;                      // merged pairs: bit h set = half h's sums were formed by its consumer
  std::map<int, View> shortcut_grad;                // op index of a bottleneck's conv1 -> gradient view of its shortcut sum
  std::map<int, int> shortcut_addact;               // ... -> index of the shortcut add whose copy the conv1 kernel absorbs
  // Wide 1x1 layers (unfused path): the weight-gradient GEMM only feeds the optimiser, so it runs on a second stream
  // beside the data-gradient GEMM of the same layer and whatever follows; joined before this function returns.
  static const bool no_aux = std::getenv("JN_NO_AUX_STREAM") != nullptr;
  bool aux_used = false;
  if (!no_aux) { int ra = ensure_aux_stream(ctx); if (ra) return ra; }
  static const bool no_red_fusion = std::getenv("JN_NO_FUSED_REDUCE") != nullptr;
  const int n_ops_b = (with_head || net.n_backbone_ops < 0) ? (int)net.ops.size() : net.n_backbone_ops;
  // JN_BWD_PROFILE=1: HIP events around the launches of every op, table on stderr (a measuring aid; use it together with
  // JN_NO_AUX_STREAM=1 so that the wide weight-gradient GEMMs are inside the brackets)
  static const bool bwd_profile = std::getenv("JN_BWD_PROFILE") != nullptr;
  std::vector<hipEvent_t> bev;
  if (bwd_profile) {
    bev.resize(n_ops_b + 1);
    for (auto& e : bev) (void)hipEventCreate(&e);
  }
  struct BwdProfileMark {        // records the event of op `i` when the loop body is left (continue / break / fall through)
    std::vector<hipEvent_t>& ev; int i; hipStream_t s;
    ~BwdProfileMark() { if (!ev.empty()) (void)hipEventRecord(ev[i], s); }
  };
  if (bwd_profile) (void)hipEventRecord(bev[n_ops_b], s);
  for (int obi = n_ops_b - 1; obi >= 0; --obi) {
    const Op& op = net.ops[obi];
    BwdProfileMark mark{bev, obi, s};
    if (op.wslot >= 0) {
      const ConvW& cw = net.convs[op.wslot];
      JN_CHECK(cw.has_bn, JN_ESTATE, "backward of BN-free conv %s inside a PAFPN", op.name.c_str());
      // the gradient of a conv that feeds a shortcut add IS the gradient of the sum: read it in place
      const auto al = g_alias.find(op.out.buf);
      const View gview = (al != g_alias.end() && al->second.first == op.out.coff) ? al->second.second : op.out;
      float* const gp_out = gptr(gview);
      const int gld_out = ld(gview);
      const long long M = (long long)N * op.out.H * op.out.W;
      double* red = net.bred + 2 * cw.stat_off;
      float* consts = net.bconsts + 3 * cw.stat_off;
      static const bool dbg_plan = std::getenv("JN_DBG_BWD_PLAN") != nullptr;
      const int hmask = red_half.count(op.wslot) ? red_half[op.wslot] : 0;
      if (dbg_plan) {
        // elements per patch that a separate bn_bwd_reduce pass re-reads (g and z: 8 bytes each); merged pairs per half
        const bool pair_halves = hmask && !cw.prefix2.empty() && net.act_dtype == JN_F32;
        const long long sep = red_done.count(op.wslot) ? 0 : pair_halves ? (M / N) * (cw.cout / 2) * (2 - ((hmask & 1) + ((hmask >> 1) & 1)))
                                                                         : (M / N) * cw.cout;
        std::fprintf(stderr, "[bwd-plan] %-34s kind %d cout %4d cin %4d M/patch %6lld stride %d acc_in %d separate-reduce elements/patch %8lld%s\n",
                     op.name.c_str(), (int)op.kind, cw.cout, cw.cin, M / N, op.stride, (int)op.acc_in, sep,
                     pair_halves ? (hmask == 3 ? " (both halves by their consumers)" : " (one half by its consumer)") : "");
      }
      if (hmask && !cw.prefix2.empty() && net.act_dtype == JN_F32) {
        // merged pair with at least one half reduced by its consumer: the other half (if any) gets its own pass, and the
        // constants are formed per half (consumer-made sums carry the moment against y, see bn_bwd_consts)
        const int h = cw.cout / 2;
        for (int half = 0; half < 2; ++half) {
          const int c0 = half * h;
          ChanTab ot = tab(op.out);
          ot.sc += c0; ot.sh += c0; ot.fl += c0;
          if (!(hmask & (1 << half)))
            launch_bn_bwd_reduce(gp_out + c0, gld_out, (const float*)ptr(op.out) + c0, net.act_dtype, ld(op.out), ot,
                                 save + 2 * (cw.stat_off + c0), h, M, red + 2 * c0, rep_stride, s, sb);
          launch_bn_bwd_consts(red + 2 * c0, rep_stride, (double)M, cw.gamma_dev + c0, cw.beta_dev + c0, save + 2 * (cw.stat_off + c0), consts + 3 * c0,
                               grad_of(ctx, cw.gamma_dev) + c0, grad_of(ctx, cw.beta_dev) + c0, h, s, sb, (hmask >> half) & 1);
        }
      } else {
        if (!red_done.count(op.wslot))
          launch_bn_bwd_reduce(gp_out, gld_out, ptr(op.out), net.act_dtype, ld(op.out), tab(op.out), save + 2 * cw.stat_off, cw.cout,
                               M, red, rep_stride, s, sb);
        launch_bn_bwd_consts(red, rep_stride, (double)M, cw.gamma_dev, cw.beta_dev, save + 2 * cw.stat_off, consts,
                             grad_of(ctx, cw.gamma_dev), grad_of(ctx, cw.beta_dev), cw.cout, s, sb, red_done.count(op.wslot) ? 1 : 0);
      }
      float* gw = grad_of(ctx, cw.w_dev);
      static const bool no_fused = std::getenv("JN_NO_FUSED_BWD") != nullptr;
      // a merged pair too wide for the fused kernel is differentiated as its two halves (independent output rows)
      const bool whole = pw_bwd_fused_supported(cw.cout, cw.cin);
      const bool halves = !whole && !cw.prefix2.empty() && 2 * cw.cout_first == cw.cout && pw_bwd_fused_supported(cw.cout_first, cw.cin);
      if (op.kind == OP_PW && net.act_dtype == JN_F32 && !no_fused && (whole || halves)) {
        const int parts = whole ? 1 : 2, pc = cw.cout / parts;
        for (int part = 0; part < parts; ++part) {
          const int c0 = part * pc;
          ChanTab ot = tab(op.out);
          ot.sc += c0; ot.sh += c0; ot.fl += c0;
          PwBwdFusedArgs fa{};
          fa.g = gp_out + c0; fa.g_ld = gld_out; fa.z = (const float*)ptr(op.out) + c0; fa.z_ld = ld(op.out); fa.ot = ot;
          fa.save = save + 2 * (cw.stat_off + c0); fa.consts = consts + 3 * c0;
          fa.x = (const float*)ptr(op.in); fa.x_ld = ld(op.in); fa.it = tab(op.in); fa.w = cw.w_dev + (size_t)c0 * cw.cin;
          fa.gx = gptr(op.in); fa.gx_ld = ld(op.in); fa.accumulate = (op.acc_in || part > 0) ? 1 : 0;
          fa.gw = gw + (size_t)c0 * cw.cin; fa.wpart = ctx->wpart; fa.M = M; fa.cout = pc; fa.cin = cw.cin; fa.sb = sb;
          const auto sg = shortcut_grad.find(obi);
          const bool folded = sg != shortcut_grad.end();
          if (folded) {                         // the shortcut add's backward left its copy to this kernel (OP_ADDACT below)
            fa.gadd = gptr(sg->second); fa.gadd_ld = ld(sg->second); fa.accumulate = 0;
          }
          // BN-backward sums of the input's producer(s) in this kernel's epilogue: it must write the FINAL gradient of
          // the view (sole reader, or the shortcut folded in) — one run, or the two halves of a CSP conv3's input
          if (parts == 1 && (!op.acc_in || folded) && !no_red_fusion && pw_bwd_fused_reduces_input(pc, cw.cin)) {
            static const bool no_half = std::getenv("JN_NO_HALF_REDUCE") != nullptr;
            const int fa_idx = folded ? shortcut_addact[obi] : -1;
            auto base_of = [&](const RedRun& r) {
              const ConvW& pcw = net.convs[r.wslot];
              const int c0 = r.half > 0 ? pcw.cout / 2 : 0;
              return net.bred + 2 * (pcw.stat_off + c0);
            };
            auto mark = [&](const RedRun& r) {
              if (r.half < 0) red_done.insert(r.wslot); else red_half[r.wslot] |= 1 << r.half;
            };
            RedRun r0, r1;
            static const bool no_red2 = std::getenv("JN_NO_RED2") != nullptr;
            // RED2: a run that is a shortcut sum -> sums of the conv behind it (its raw output and table instead of the input's)
            auto red2 = [&](int coff, int C, int fidx) {
              if (no_red2) return false;
              int add = -1;
              const int conv = shortcut_sum_conv(net, obi, op.in.buf, coff, C, fidx, &add);
              if (conv < 0) return false;
              const Op& co = net.ops[conv];
              const ConvW& pcw = net.convs[co.wslot];
              const ChanTab zt = tab(co.out);
              fa.red2_z = (const float*)ptr(co.out); fa.red2_ld = ld(co.out); fa.red2_sc = zt.sc; fa.red2_sh = zt.sh;
              fa.red_in = net.bred + 2 * pcw.stat_off; fa.red_rep_stride = rep_stride;
              red_done.insert(co.wslot);
              return true;
            };
            if (red_segment(net, obi, op.in.buf, op.in.coff, op.in.C, fa_idx, r0) && (r0.half < 0 || !no_half)) {
              fa.red_in = base_of(r0); fa.red_rep_stride = rep_stride; mark(r0);
            } else if (red2(op.in.coff, op.in.C, fa_idx)) {
              fa.red_split = op.in.C;
            } else if (!no_half && op.in.C % 32 == 0 && !folded) {
              const int hC = op.in.C / 2;
              bool ok0 = red_segment(net, obi, op.in.buf, op.in.coff, hC, -1, r0);
              const bool ok1 = red_segment(net, obi, op.in.buf, op.in.coff + hC, hC, -1, r1);
              if (!ok0 && red2(op.in.coff, hC, -1)) {        // [shortcut sum | conv2 half]: the input of a CSP's conv3
                fa.red_split = hC;
                if (ok1) { fa.red_in2 = base_of(r1); mark(r1); }
              } else
              if (ok0 || ok1) {
                fa.red_rep_stride = rep_stride; fa.red_split = hC;
                if (ok0) { fa.red_in = base_of(r0); mark(r0); }
                if (ok1) { fa.red_in2 = base_of(r1); mark(r1); }
                if (!ok1) fa.red_in2 = nullptr;
                // (a lone second run still needs the split: red_in stays null, red_in2 set)
                if (!ok0 && ok1) fa.red_in = nullptr;
              }
            }
          }
          launch_pw_bwd_fused(fa, s);
        }
        continue;
      }
      static const bool no_fused_dw = std::getenv("JN_NO_FUSED_DW") != nullptr;
      if (op.kind == OP_DW && net.act_dtype == JN_F32 && !no_fused && !no_fused_dw &&
          dw_bwd_fused_supported(cw.cout, op.in.H, op.in.W, op.out.H, op.out.W, op.stride)) {
        DwBwdFusedArgs fa{};
        fa.g = gp_out; fa.g_ld = gld_out; fa.z = (const float*)ptr(op.out); fa.z_ld = ld(op.out); fa.ot = tab(op.out);
        fa.save = save + 2 * cw.stat_off; fa.consts = consts;
        fa.x = (const float*)ptr(op.in); fa.x_ld = ld(op.in); fa.it = tab(op.in); fa.w = cw.w_dev;
        fa.gin = gptr(op.in); fa.gin_ld = ld(op.in); fa.accumulate = op.acc_in ? 1 : 0; fa.gw = gw; fa.wpart = ctx->wpart;
        fa.C = cw.cout; fa.H = op.in.H; fa.W = op.in.W; fa.OH = op.out.H; fa.OW = op.out.W; fa.N = N; fa.stride = op.stride;
        fa.sb = sb;
        // stride 2 (round 4): the owner-staged variant of the kernel (every thread stages the 2 x 2 input block it
        // differentiates and keeps the raw values); round 3's variant re-read the input inside the gradient loop and lost
        // more than the separate pass costs.  A view that only the network's outside could also write (an FPN output) and
        // that gets no outside gradient in this backward (fpn_zero) is an ordinary single-consumer output: this kernel
        // then WRITES its gradient (the buffer holds zeros) and forms the producer's sums like for any other.
        static const bool no_s2_red = std::getenv("JN_NO_S2_RED") != nullptr;
        if (!no_red_fusion && (op.stride == 1 || !no_s2_red)) {
          bool ext_zero = false;
          for (int i = 0; i < 3; ++i) ext_zero = ext_zero || (((fpn_zero >> i) & 1) && views_overlap(op.in, net.fpn[i]));
          if (!op.acc_in || ext_zero) {
            const int prod = sole_producer(net, obi, fpn_zero);
            if (prod >= 0) {
              const ConvW& pcw = net.convs[net.ops[prod].wslot];
              fa.red_in = net.bred + 2 * pcw.stat_off; fa.red_rep_stride = rep_stride;
              fa.accumulate = 0;                      // sole reader: nothing but the (zero) outside seed was there before
              red_done.insert(net.ops[prod].wslot);
            }
          }
        }
        launch_dw_bwd_fused(fa, s);
        continue;
      }
      if (op.kind == OP_STEM && net.act_dtype == JN_F32 && !no_fused) {
        StemArgs a{ss.src, ss.positions, ss.pos_stride, ss.sample_stride, ss.chan_stride, ss.row_stride, net.P, N, cw.cout,
                   cw.w_dev, nullptr, 0, JN_F32, nullptr, 0, nullptr, 0};
        launch_stem_bwd_weight(a, gp_out, gld_out, gw, ctx->wpart, s, sb, (const float*)ptr(op.out), ld(op.out),
                               tab(op.out), save + 2 * cw.stat_off, consts);
        continue;
      }
      launch_bn_bwd_gz(gp_out, gld_out, ptr(op.out), net.act_dtype, ld(op.out), tab(op.out), save + 2 * cw.stat_off, consts,
                       cw.cout, M, s, sb);
      if (op.kind == OP_PW) {
        ConvArgs a{};
        a.in = gp_out; a.in_ld = gld_out; a.in_dtype = JN_F32; a.itab = ident; a.w = cw.w_dev; a.bias = nullptr;
        a.out = gptr(op.in); a.out_ld = ld(op.in); a.out_dtype = JN_F32; a.bf16_mfma = net.act_dtype == JN_BF16;
        a.N = N; a.H = op.out.H; a.W = op.out.W; a.OH = op.out.H; a.OW = op.out.W;
        a.cin = cw.cout; a.cout = cw.cin; a.stride = 1; a.act = ACT_NONE;
        a.accumulate = op.acc_in ? 1 : 0; a.w_transposed = 1; a.in_identity = 1;
        a.n_slots = nsl; a.in_slot_stride = sb.grad; a.out_slot_stride = sb.grad; a.tab_slot_stride = 0;
        hipStream_t ws = s;
        if (!no_aux && cw.cout >= 128 && cw.cin >= 128) {      // the wide kernel: plain atomics on gw, no shared scratch
          JN_HIP(hipEventRecord(ctx->aux_fork, s));           // g_z is complete
          JN_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
          ws = ctx->aux_stream;
          aux_used = true;
        }
        // wide layers, fp32: the data gradient on the bf16 pipe at fp32 accuracy (pw_x3_kernel over the slots, transposed
        // weight split on the way: kernels_pwxs.hip); JN_NO_PW_X3_BWD=1 keeps pw_dir_kernel<WT> on the fp32 pipe
        const bool no_x3_bwd = std::getenv("JN_NO_PW_X3_BWD") != nullptr;     // read per launch: a test flips it
        bool x3_done = false;
        if (!no_x3_bwd && net.act_dtype == JN_F32 && !op.acc_in && ctx->params_x3t && pw_x3_bwd_data_supported(cw.cout, cw.cin) &&
            (cw.w_dev - ctx->params) % 8 == 0) {
          x3_done = launch_pw_x3_bwd_data(a, cw.w_dev, ctx->params_x3t + 3 * (cw.w_dev - ctx->params), s) == 0;
        }
        if (!x3_done) launch_pw(a, s);
        launch_pw_bwd_weight(gp_out, gld_out, ptr(op.in), net.act_dtype, ld(op.in), tab(op.in), gw, ctx->wpart, M, cw.cout,
                             cw.cin, ws, sb);
      } else if (op.kind == OP_DW) {
        launch_dw_bwd_data(gp_out, gld_out, cw.w_dev, gptr(op.in), ld(op.in), cw.cout, op.in.H, op.in.W, op.out.H,
                           op.out.W, N, op.stride, op.acc_in ? 1 : 0, s, sb);
        launch_dw_bwd_weight(gp_out, gld_out, ptr(op.in), net.act_dtype, ld(op.in), tab(op.in), gw, ctx->wpart, cw.cout, op.in.H,
                             op.in.W, op.out.H, op.out.W, N, op.stride, s, sb);
      } else if (op.kind == OP_STEM) {
        StemArgs a{ss.src, ss.positions, ss.pos_stride, ss.sample_stride, ss.chan_stride, ss.row_stride, net.P, N, cw.cout,
                   cw.w_dev, nullptr, 0, JN_F32, nullptr, 0, nullptr, 0};
        launch_stem_bwd_weight(a, gp_out, gld_out, gw, ctx->wpart, s, sb);
      } else if (op.kind == OP_CONV3) {
        // dense 3x3 (non-depthwise patch encoders, e.g. yolox-s): stride 1 = the forward kernel over g_z with
        // mirrored taps and the transposed weight; stride 2 = one MFMA tile loop per input-pixel parity class
        int rc3 = 0;
        hipStream_t ws3 = s;
        if (!no_aux) {                     // the 9-tap weight gradient (plain atomics on gw) beside the data gradient
          JN_HIP(hipEventRecord(ctx->aux_fork, s));
          JN_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
          ws3 = ctx->aux_stream;
          aux_used = true;
        }
        if (op.stride == 1) {
          ConvArgs a{};
          a.in = gp_out; a.in_ld = gld_out; a.in_dtype = JN_F32; a.itab = ident; a.w = cw.w_dev; a.bias = nullptr;
          a.out = gptr(op.in); a.out_ld = ld(op.in); a.out_dtype = JN_F32;
          a.N = N; a.H = op.out.H; a.W = op.out.W; a.OH = op.in.H; a.OW = op.in.W;
          a.cin = cw.cout; a.cout = cw.cin; a.stride = 1; a.act = ACT_NONE;
          a.accumulate = op.acc_in ? 1 : 0; a.w_transposed = 1; a.in_identity = 1;
          a.n_slots = nsl; a.in_slot_stride = sb.grad; a.out_slot_stride = sb.grad;
          rc3 = launch_conv3(a, s);
        } else {
          rc3 = launch_conv3_bwd_data_s2(gp_out, gld_out, cw.w_dev, gptr(op.in), ld(op.in), op.in.H, op.in.W, op.out.H,
                                         op.out.W, cw.cout, cw.cin, N, op.acc_in ? 1 : 0, s, sb);
        }
        JN_CHECK(rc3 == 0, JN_ESTATE, "backward of dense 3x3 conv %s: unsupported shape", op.name.c_str());
        launch_conv3_bwd_weight(gp_out, gld_out, ptr(op.in), net.act_dtype, ld(op.in), tab(op.in), gw, op.in.H, op.in.W,
                                op.out.H, op.out.W, cw.cout, cw.cin, N, op.stride, ws3, sb);
      } else {
        set_error("backward of op %s is not implemented", op.name.c_str());
        return JN_ESTATE;
      }
      continue;
    }
    switch (op.kind) {
      case OP_ADDACT: {
        const long long M = (long long)N * op.out.H * op.out.W;
        if (op.acc_in) launch_grad_copy(gptr(op.out), ld(op.out), gptr(op.in), ld(op.in), op.out.C, M, 1, s, sb);
        else g_alias[op.in.buf] = std::make_pair(op.in.coff, op.out);       // sole consumer: no copy, see above
        // The shortcut branch: g[res] += g[sum].  When the only other reader of `res` is the bottleneck's first 1x1 conv
        // and that layer takes the fused backward kernel, the kernel adds g[sum] while it writes its data gradient (one
        // read of g[sum] instead of a copy pass — read, read-modify-write — over the largest 16 / 32-channel maps)
        {
          static const bool no_fold = std::getenv("JN_NO_SHORTCUT_FOLD") != nullptr;
          static const bool no_fused2 = std::getenv("JN_NO_FUSED_BWD") != nullptr;
          int conv1 = -1, readers = 0;
          for (int j = 0; j < n_ops_b; ++j) {
            const Op& o = net.ops[j];
            if (j != obi && (views_overlap(o.in, op.res) || views_overlap(o.res, op.res))) {
              ++readers;
              if (o.kind == OP_PW && o.in.buf == op.res.buf && o.in.coff == op.res.coff && o.in.C == op.res.C && j < obi) conv1 = j;
            }
          }
          bool fold = !no_fold && !no_fused2 && !op.acc_res && readers == 1 && conv1 >= 0 && net.act_dtype == JN_F32;
          if (fold) {
            const Op& c1 = net.ops[conv1];
            const ConvW& cw1 = net.convs[c1.wslot];
            fold = c1.acc_in && cw1.prefix2.empty() && pw_bwd_fused_supported(cw1.cout, cw1.cin);
          }
          if (fold) {
            // g[sum] must still hold the gradient when that kernel runs: the conv that feeds the add reads it in place
            // (g_alias above) before — fine for the fused 1x1 kernel, which leaves it alone, not for the unfused paths,
            // whose bn_bwd_gz turns it into g_z IN PLACE (dense 3x3 bottlenecks of the non-depthwise encoders)
            bool intact = false;
            for (int j = 0; j < obi; ++j) {
              const Op& o = net.ops[j];
              if (o.wslot >= 0 && o.out.buf == op.in.buf && o.out.coff == op.in.coff && o.out.C == op.in.C) {
                const ConvW& cwp = net.convs[o.wslot];
                intact = o.kind == OP_PW && cwp.prefix2.empty() && pw_bwd_fused_supported(cwp.cout, cwp.cin);
              }
            }
            fold = intact;
          }
          if (fold) { shortcut_grad[conv1] = op.out; shortcut_addact[conv1] = obi; }
          else launch_grad_copy(gptr(op.out), ld(op.out), gptr(op.res), ld(op.res), op.out.C, M, op.acc_res ? 1 : 0, s, sb);
        }
        break;
      }
      case OP_SPP: {
        const View full = net_full_view(net, op.out.buf);
        launch_spp_bwd(ptr(full), net.act_dtype, gptr(full), ld(op.out), op.in.C, op.in.H, op.in.W, N, tab(op.in), s, sb);
        break;
      }
      case OP_UPSAMPLE:
        launch_upsample_bwd(gptr(op.out), ld(op.out), gptr(op.in), ld(op.in), op.in.C, op.in.H, op.in.W, N,
                            op.acc_in ? 1 : 0, s, sb);
        break;
      default: break;
    }
  }
  if (aux_used) {
    JN_HIP(hipEventRecord(ctx->aux_join, ctx->aux_stream));
    JN_HIP(hipStreamWaitEvent(s, ctx->aux_join, 0));
  }
  if (bwd_profile) {
    (void)hipStreamSynchronize(s);
    static const char* kn[] = {"stem", "pw", "dw", "conv3", "spp", "upsample", "addact", "pred"};
    double tot_us = 0, tot_b = 0;
    fprintf(stderr, "# backward profile: net %d, N=%d patches x %d steps; bytes = g_out + z_out + x read, g_in written (+ read when accumulated)\n", ni, N, nsl);
    for (int obi = n_ops_b - 1; obi >= 0; --obi) {
      const Op& op = net.ops[obi];
      float ms = 0;
      (void)hipEventElapsedTime(&ms, bev[obi + 1], bev[obi]);
      const double in_e = op.kind == OP_STEM ? 0.0 : (double)op.in.H * op.in.W * op.in.C, out_e = (double)op.out.H * op.out.W * op.out.C;
      double elems;
      if (op.wslot >= 0) elems = 2.0 * out_e + (op.kind == OP_STEM ? 3.0 * net.P * net.P : 2.0 * in_e + (op.acc_in ? in_e : 0.0));
      else if (op.kind == OP_ADDACT) elems = 3.0 * out_e;                 // g read, two destinations
      else if (op.kind == OP_SPP) elems = 8.0 * in_e;
      else elems = in_e + out_e;
      const double bytes = elems * 4.0 * N * nsl;
      tot_us += ms * 1e3; tot_b += bytes;
      fprintf(stderr, "%-8s %-44s in %3dx%3dx%3d out %3dx%3dx%3d s%d acc %d  %8.1f us  %8.1f MB  %6.0f GB/s\n", kn[op.kind], op.name.c_str(),
              op.in.H, op.in.W, op.in.C, op.out.H, op.out.W, op.out.C, op.stride, (int)op.acc_in, ms * 1e3, bytes / 1e6, bytes / (ms * 1e-3) / 1e9);
    }
    fprintf(stderr, "# total %.1f us, %.1f MB, %.0f GB/s\n", tot_us, tot_b / 1e6, tot_b / (tot_us * 1e-6) / 1e9);
    for (auto& e : bev) (void)hipEventDestroy(e);
  }
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// embed_fpn (src/models/gpt.py:294-306, 382) on the last FPN map of the encoder for N patches:
// 1x1 conv + ReLU, then the split-K partial sums of the Linear (finished by the consumer).
static int run_embed_fpn(jn_ctx* ctx, int N, int slot, float* e_buf, const int* skip_flag, int skip_when,
                         hipStream_t s) {
  if (!e_buf) e_buf = ctx->efpn_act;
  const Net& net = ctx->nets[ctx->enc_net];
  jn_ctx& x = *ctx;
  const int C = ctx->cfg.n_embd, MB = ctx->cfg.max_batch;
  const View& f = net.fpn[2];
  ConvArgs a{};
  a.in = view_ptr(net, slot, MB, f); a.in_ld = net.bufs[f.buf].C; a.in_dtype = net.act_dtype; a.itab = view_tab(net, slot, f);
  a.out_dtype = JN_F32; a.bf16_mfma = net.act_dtype == JN_BF16;
  a.w = ctx->gpt.efpn_w; a.bias = nullptr; a.out = e_buf; a.out_ld = C;
  a.N = N; a.H = f.H; a.W = f.W; a.OH = f.H; a.OW = f.W; a.cin = f.C; a.cout = C; a.stride = 1; a.act = ACT_RELU;
  a.skip_flag = skip_flag; a.skip_when = skip_when;
  launch_pw(a, s);
  launch_efpn_linear(e_buf, ctx->gpt.efpn_lin_wt, x.emb_part, N, f.H * f.W * C, C, x.KS, skip_flag, skip_when, s);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_backbone_forward(jn_ctx* ctx, int net, const float* patches_dev, int N, int train, float* fpn0_dev,
                        float* fpn1_dev, float* fpn2_dev, void* stream) {
  JN_CHECK(ctx && patches_dev, JN_EINVAL, "jn_backbone_forward: null argument");
  JN_CHECK(net >= 0 && net < 2 && ctx->has_net[net], JN_EINVAL, "network %d is not part of this context", net);
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(N >= 1 && N <= ctx->cfg.max_batch, JN_EINVAL, "N=%d exceeds max_batch=%d", N, ctx->cfg.max_batch);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int P = ctx->cfg.patch_size;
  StemSrc ss{patches_dev, nullptr, 3LL * P * P, (long long)P * P, P};
  int rc = run_net(ctx, net, N, ss, 0, train ? 1 : 0, nullptr, 0, s);
  if (rc) return rc;
  float* outs[3] = {fpn0_dev, fpn1_dev, fpn2_dev};
  const Net& n = ctx->nets[net];
  for (int i = 0; i < 3; ++i) {
    if (!outs[i]) continue;
    const View& f = n.fpn[i];
    launch_nhwc_to_nchw(view_ptr(n, 0, ctx->cfg.max_batch, f), n.act_dtype, n.bufs[f.buf].C, view_tab(n, 0, f), outs[i], f.C,
                        f.H * f.W, N, s);
  }
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// finishes the split-K sums: out[n][c] = bias[c] + sum_ks part[n][ks][c]
__global__ void emb_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                  long long out_stride, int N, int KS, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  float s = bias[c];
  for (int k = 0; k < KS; ++k) s += part[((long long)n * KS + k) * C + c];
  out[(long long)n * out_stride + c] = s;
}

int jn_embed_patches(jn_ctx* ctx, const float* patches_dev, int N, float* out_dev, void* stream) {
  JN_CHECK(ctx && patches_dev && out_dev, JN_EINVAL, "jn_embed_patches: null argument");
  JN_CHECK(!ctx->cfg.no_patch_emb, JN_ESTATE, "context was created with no_patch_emb");
  int rc = jn_backbone_forward(ctx, ctx->enc_net, patches_dev, N, 0, nullptr, nullptr, nullptr, stream);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = run_embed_fpn(ctx, N, 0, nullptr, nullptr, 0, s))) return rc;
  const int C = ctx->cfg.n_embd;
  hipLaunchKernelGGL(emb_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, ctx->emb_part,
                     ctx->gpt.efpn_lin_b, out_dev, (long long)C, N, ctx->KS, C);
  JN_HIP(hipGetLastError());
  return JN_OK;
}


static void fill_gpt_weights(const jn_ctx* ctx, GptStepArgs& a) {
  const jn_config& c = ctx->cfg;
  const GptW& g = ctx->gpt;
  a.C = c.n_embd; a.n_head = c.n_head; a.n_layer = c.n_layer; a.nA = c.n_actions; a.Tmax = c.block_size + 1;
  a.use_pos_emb = c.use_pos_emb; a.no_patch_emb = c.no_patch_emb; a.concat_emb = c.concat_emb;
  a.dec_pos_enc = c.decoder_pos_encoding; a.n_parts = n_parts(c);
  a.pe2_ch = (int)std::ceil(c.n_embd / 4.0) * 2;
  a.wte = g.wte; a.wpe = g.wpe; a.embed_class = g.embed_class; a.proj_wt = g.proj_wt; a.proj_b = g.proj_b;
  a.pos1d = g.pos1d; a.pe2 = g.pos2d_col; a.head_wt = g.head_wt; a.lnf_w = g.lnf_w; a.lnf_b = g.lnf_b;
  a.layers = ctx->layers_dev; a.emb_part = ctx->emb_part; a.KS = ctx->KS; a.efpn_lin_b = g.efpn_lin_b;
  a.kcache = ctx->kcache; a.vcache = ctx->vcache; a.prev_action = ctx->prev_action; a.cache_len = ctx->cache_len;
  a.n_done = ctx->n_done;
}

int jn_gpt_forward(jn_ctx* ctx, const float* patches_dev, const int64_t* actions_dev, const int64_t* classes_dev,
                   const int64_t* positions_dev, const float* prev_emb_dev, int B, int T, int Tp, float* logits_dev,
                   float* final_emb_dev, void* stream) {
  JN_CHECK(ctx && actions_dev, JN_EINVAL, "jn_gpt_forward: null argument");
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  const jn_config& c = ctx->cfg;
  JN_CHECK(B >= 1 && B <= c.max_batch, JN_EINVAL, "B=%d exceeds max_batch=%d", B, c.max_batch);
  // gpt.py:514-518
  JN_CHECK(T >= 1 && T <= c.block_size, JN_EINVAL, "Cannot forward sequence of length %d, block size is only %d", T,
           c.block_size);
  JN_CHECK(!c.use_pos_emb || positions_dev, JN_EINVAL, "positions are required when use_pos_emb is set");
  JN_CHECK(c.no_patch_emb || patches_dev, JN_EINVAL, "patches are required unless no_patch_emb is set");
  JN_CHECK(!prev_emb_dev || (Tp >= 1 && Tp + 1 <= c.block_size + 1), JN_EINVAL, "prev_embeddings length %d out of range", Tp);
  JN_HIP(hipSetDevice(c.device));
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_embd, P = c.patch_size, nA = c.n_actions;
  const int n_new = prev_emb_dev ? 1 : T, i0 = prev_emb_dev ? T - 1 : 0;
  const int L = prev_emb_dev ? Tp + 1 : T + 1;
  int rc;
  if (!c.no_patch_emb) {
    if (!ctx->tok_emb) {
      if ((rc = dev_alloc(ctx, &ctx->tok_emb, (size_t)c.max_batch * (c.block_size + 1) * C))) return rc;
    }
    for (int i = 0; i < n_new; ++i) {
      StemSrc ss{patches_dev + (size_t)(i0 + i) * 3 * P * P, nullptr, (long long)T * 3 * P * P, (long long)P * P, P};
      if ((rc = run_net(ctx, ctx->enc_net, B, ss, 0, 0, nullptr, 0, s))) return rc;
      if ((rc = run_embed_fpn(ctx, B, 0, nullptr, nullptr, 0, s))) return rc;
      hipLaunchKernelGGL(emb_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, ctx->emb_part, ctx->gpt.efpn_lin_b,
                         ctx->tok_emb + (size_t)i * C, (long long)n_new * C, B, ctx->KS, C);
    }
  }
  JN_HIP(hipMemsetAsync(ctx->cache_len, 0, (size_t)B * sizeof(int32_t), s));
  GptStepArgs a{};
  fill_gpt_weights(ctx, a);
  a.classes = classes_dev;
  a.B = B; a.T = L; a.emb_stride = L; a.out.final_emb = final_emb_dev; a.logits_stride = (L - 1) * nA;
  for (int tok = 0; tok < L; ++tok) {
    a.step = tok;
    a.logits_rows = (tok >= 1 && logits_dev) ? logits_dev + (size_t)(tok - 1) * nA : nullptr;
    if (prev_emb_dev && tok < Tp) {
      a.src_mode = GPT_SRC_GIVEN; a.given_emb = prev_emb_dev; a.given_stride = Tp; a.given_index = tok;
    } else if (!prev_emb_dev && tok == 0) {
      a.src_mode = GPT_SRC_CLASS;
    } else {
      const int i = prev_emb_dev ? 0 : tok - 1;           // index among the new tokens
      a.src_mode = GPT_SRC_TEACH;
      a.t_actions = actions_dev; a.t_positions = positions_dev; a.t_stride = T; a.t_index = i0 + i;
      a.pos_index = prev_emb_dev ? 0 : i;                 // recurrent tokens always get position 0 (gpt.py:431-449)
      a.tok_emb = ctx->tok_emb; a.tok_emb_stride = n_new; a.tok_emb_index = i;
    }
    launch_gpt_step(a, s);
  }
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_read_tensor(jn_ctx* ctx, const char* name, float* host_out, size_t numel) {
  JN_CHECK(ctx && name && host_out, JN_EINVAL, "jn_read_tensor: null argument");
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  const std::string nm(name);
  for (int ni = 0; ni < 2; ++ni) {
    if (!ctx->has_net[ni]) continue;
    for (const ConvW& cw : ctx->nets[ni].convs) {
      if (!cw.has_bn) continue;
      // a merged pair answers for both of its modules: [0, cout_first) and [cout_first, cout)
      const bool first = nm.compare(0, cw.prefix.size(), cw.prefix) == 0 && nm.size() > cw.prefix.size() && nm[cw.prefix.size()] == '.';
      const bool second = !cw.prefix2.empty() && nm.compare(0, cw.prefix2.size(), cw.prefix2) == 0 &&
                          nm.size() > cw.prefix2.size() && nm[cw.prefix2.size()] == '.';
      if (!first && !second) continue;
      const std::string leaf = nm.substr(first ? cw.prefix.size() : cw.prefix2.size());
      const float* src = leaf == ".bn.running_mean" ? cw.rmean_dev : leaf == ".bn.running_var" ? cw.rvar_dev : nullptr;
      if (!src) continue;
      int n_here = cw.cout;
      if (!cw.prefix2.empty()) { n_here = first ? cw.cout_first : cw.cout - cw.cout_first; if (second) src += cw.cout_first; }
      JN_CHECK(numel == (size_t)n_here, JN_EINVAL, "'%s' has %d elements, not %zu", name, n_here, numel);
      JN_HIP(hipDeviceSynchronize());
      JN_HIP(hipMemcpy(host_out, src, numel * sizeof(float), hipMemcpyDeviceToHost));
      return JN_OK;
    }
  }
  set_error("jn_read_tensor: '%s' is not a tensor the engine updates", name);
  return JN_ENOTFOUND;
}

int jn_zero_grad(jn_ctx* ctx, void* stream) {
  JN_CHECK(ctx && ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  int rc = ensure_train_state(ctx);
  if (rc) return rc;
  JN_HIP(hipMemsetAsync(ctx->grads, 0, ctx->arena_size * sizeof(float), (hipStream_t)stream));
  return JN_OK;
}

int jn_backbone_backward(jn_ctx* ctx, int net, const float* patches_dev, int N, const float* g0_dev,
                         const float* g1_dev, const float* g2_dev, void* stream) {
  JN_CHECK(ctx && patches_dev, JN_EINVAL, "jn_backbone_backward: null argument");
  JN_CHECK(net >= 0 && net < 2 && ctx->has_net[net], JN_EINVAL, "network %d is not part of this context", net);
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(N >= 1 && N <= ctx->cfg.max_batch, JN_EINVAL, "N=%d exceeds max_batch=%d", N, ctx->cfg.max_batch);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  int rc = ensure_train_state(ctx);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  Net& n = ctx->nets[net];
  const int P = ctx->cfg.patch_size, MB = ctx->cfg.max_batch;
  const float* gs[3] = {g0_dev, g1_dev, g2_dev};
  for (int i = 0; i < 3; ++i) {
    const View& f = n.fpn[i];
    float* gp = n.gact + n.buf_off[f.buf] * (size_t)MB + f.coff;
    if (gs[i]) {
      launch_nchw_to_nhwc_grad(gs[i], gp, n.bufs[f.buf].C, f.C, f.H * f.W, N, 0, s);
    } else {
      JN_CHECK(n.bufs[f.buf].C == f.C, JN_ESTATE, "fpn output is a slice");
      JN_HIP(hipMemsetAsync(gp, 0, (size_t)N * f.H * f.W * f.C * sizeof(float), s));
    }
  }
  StemSrc ss{patches_dev, nullptr, 3LL * P * P, (long long)P * P, P};
  return run_net_backward(ctx, net, N, ss, 0, s);
}

int jn_read_grad(jn_ctx* ctx, const char* name, float* host_out, size_t numel) {
  JN_CHECK(ctx && name && host_out, JN_EINVAL, "jn_read_grad: null argument");
  JN_CHECK(ctx->grads, JN_ESTATE, "no gradient has been computed yet");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  auto it = ctx->seg_index.find(name);
  JN_CHECK(it != ctx->seg_index.end(), JN_ENOTFOUND, "jn_read_grad: '%s' is not a trainable tensor", name);
  const ParamSeg& sg = ctx->segs[it->second];
  JN_CHECK(sg.numel == numel, JN_EINVAL, "'%s' has %zu elements, not %zu", name, sg.numel, numel);
  std::vector<float> packed(numel);
  JN_HIP(hipDeviceSynchronize());
  JN_HIP(hipMemcpy(packed.data(), ctx->grads + sg.off, numel * sizeof(float), hipMemcpyDeviceToHost));
  const std::vector<float> t = unpack_param(sg, packed);
  std::memcpy(host_out, t.data(), numel * sizeof(float));
  return JN_OK;
}

static int detect_impl(jn_ctx* ctx, const StemSrc& ss, int N, float* boxes_dev, int32_t* counts_dev, float* raw_dev,
                       const int* skip_flag, int skip_when, hipStream_t s) {
  Net& net = ctx->nets[JN_NET_DETECTOR];
  int rc;
  if (!ctx->det_raw)
    if ((rc = dev_alloc(ctx, &ctx->det_raw, (size_t)ctx->cfg.max_batch * net.n_anchors * 6))) return rc;
  if ((rc = run_net(ctx, JN_NET_DETECTOR, N, ss, 0, 0, skip_flag, skip_when, s, true))) return rc;
  if (raw_dev)
    JN_HIP(hipMemcpyAsync(raw_dev, ctx->det_raw, (size_t)N * net.n_anchors * 6 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (boxes_dev && counts_dev)
    launch_postprocess(ctx->det_raw, net.n_anchors, N, ctx->cfg.det_conf_threshold, ctx->cfg.det_nms_threshold,
                       (float)(ctx->cfg.patch_size - 1), boxes_dev, counts_dev, ctx->cfg.max_det_per_patch, s);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_detect(jn_ctx* ctx, const float* patches_dev, int N, float* boxes_dev, int32_t* counts_dev, float* raw_dev,
              void* stream) {
  JN_CHECK(ctx && patches_dev, JN_EINVAL, "jn_detect: null argument");
  JN_CHECK(ctx->has_net[JN_NET_DETECTOR], JN_ESTATE, "context was created without a detector");
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(N >= 1 && N <= ctx->cfg.max_batch, JN_EINVAL, "N=%d exceeds max_batch=%d", N, ctx->cfg.max_batch);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  const int P = ctx->cfg.patch_size;
  StemSrc ss{patches_dev, nullptr, 3LL * P * P, (long long)P * P, P};
  return detect_impl(ctx, ss, N, boxes_dev, counts_dev, raw_dev, nullptr, 0, (hipStream_t)stream);
}

// xyxy (class, x1, y1, x2, y2) -> (class, cx, cy, w, h), src/models/yolox.py:59-60
__global__ void labels_to_cxcywh_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* s = in + 5 * i;
  float* d = out + 5 * i;
  d[0] = s[0]; d[1] = 0.5f * (s[1] + s[3]); d[2] = 0.5f * (s[2] + s[4]); d[3] = s[3] - s[1]; d[4] = s[4] - s[2];
}

// ---- detector training: NeedleYOLOX.forward(patches, targets) (src/models/yolox.py:24-91) in two halves ------------
// A "pass" is one train-mode run of PAFPN + head over up to max_batch patches whose activations stay resident in a
// workspace slot of their own until the backward: slot det_slot_base + pass.  With a separate patch encoder the
// detector net's slot 0 is the eval workspace and the passes start at 1; when the detector's own PAFPN encodes the
// patches (no gpt_backbone) a train-mode rollout owns slots 1 .. T <= block_size of the SAME net, so the passes start
// behind them — the rollout's backward may then run after the detector pass, the reference's statement order
// (src/reinforce.py:326-341).
static inline int det_slot_base(const jn_ctx* ctx) { return ctx->enc_net == JN_NET_DETECTOR ? ctx->cfg.block_size + 1 : 1; }

__global__ void det_scale_kernel(const float* __restrict__ fwd_scale, const float* __restrict__ dloss, float host_scale,
                                 float* __restrict__ out) {
  out[0] = fwd_scale[0] * (dloss ? dloss[0] : 1.0f) * host_scale;
}

static int detector_forward_impl(jn_ctx* ctx, const float* patches_dev, int N, const float* targets_dev, int nb, int pass,
                                 int n_pass, float loss_scale, float* metrics_dev, hipStream_t s) {
  JN_CHECK(ctx && patches_dev && targets_dev && metrics_dev, JN_EINVAL, "detector training pass: null argument");
  JN_CHECK(ctx->has_net[JN_NET_DETECTOR], JN_ESTATE, "context was created without a detector");
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(N >= 1 && N <= ctx->cfg.max_batch, JN_EINVAL, "N=%d exceeds max_batch=%d", N, ctx->cfg.max_batch);
  JN_CHECK(nb >= 1, JN_EINVAL, "targets need at least one (padding) row per patch");
  JN_CHECK(n_pass >= 1 && n_pass <= 64 && pass >= 0 && pass < n_pass, JN_EINVAL, "detector pass %d of %d", pass, n_pass);
  JN_CHECK(ctx->cfg.act_dtype == JN_F32, JN_ESTATE, "training needs act_dtype = fp32 (bf16 is the inference mode)");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  Net& net = ctx->nets[JN_NET_DETECTOR];
  const int MB = ctx->cfg.max_batch, A = net.n_anchors, P = ctx->cfg.patch_size;
  const int slot = det_slot_base(ctx) + pass;
  int rc;
  if ((rc = ensure_slots(ctx, net, det_slot_base(ctx) + n_pass))) return rc;
  if ((rc = ensure_train_state(ctx, ctx->enc_net == JN_NET_DETECTOR ? std::max(1, ctx->nets[ctx->enc_net].g_slots) : 1))) return rc;
  if (!ctx->det_logits) {
    if ((rc = dev_alloc(ctx, &ctx->det_logits, (size_t)MB * A * 6))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->det_bwd_scale, (size_t)4))) return rc;
  }
  if ((int)ctx->det_pass.size() < n_pass) ctx->det_pass.resize(n_pass);
  jn_ctx::DetPass& dp = ctx->det_pass[pass];
  dp.valid = false;
  if (!dp.dlogits) {
    if ((rc = dev_alloc(ctx, &dp.dlogits, (size_t)MB * A * 6))) return rc;
    if ((rc = dev_alloc(ctx, &dp.acc, (size_t)16))) return rc;
  }
  if (!ctx->det_labels || ctx->det_labels_rows < (size_t)N * nb) {
    if ((rc = dev_alloc(ctx, &ctx->det_labels, (size_t)MB * nb * 5))) return rc;
    ctx->det_labels_rows = (size_t)MB * nb;
  }
  hipLaunchKernelGGL(labels_to_cxcywh_kernel, dim3((N * nb + 255) / 256), dim3(256), 0, s, targets_dev, ctx->det_labels, N * nb);
  StemSrc ss{patches_dev, nullptr, 3LL * P * P, (long long)P * P, P};
  if ((rc = run_net(ctx, JN_NET_DETECTOR, N, ss, slot, 1, nullptr, 0, s, true))) return rc;
  DetGeom geo{};
  geo.A = A;
  for (const Op& op : net.ops) {
    if (op.kind != OP_PRED) continue;
    geo.a0[op.level] = op.anchor0; geo.H[op.level] = op.in.H; geo.W[op.level] = op.in.W; geo.stride[op.level] = op.stride;
  }
  launch_yolox_loss(ctx->det_logits, ctx->det_labels, N, nb, geo, dp.dlogits, dp.acc, 1, loss_scale, metrics_dev, dp.acc + 8, s);
  JN_HIP(hipGetLastError());
  dp.patches = patches_dev; dp.N = N; dp.valid = true;
  return JN_OK;
}

// backward of pass `pass`: d loss / d raw (left by the forward, scaled by scale_dev[0]) through the predictors, the head
// and the PAFPN; parameter gradients ACCUMULATE in the arena
static int detector_backward_impl(jn_ctx* ctx, int pass, const float* scale_dev, hipStream_t s) {
  JN_CHECK(pass >= 0 && pass < (int)ctx->det_pass.size() && ctx->det_pass[pass].valid, JN_ESTATE,
           "detector backward: pass %d has no forward to differentiate (none ran, or a later pass overwrote its activations)", pass);
  Net& net = ctx->nets[JN_NET_DETECTOR];
  const jn_ctx::DetPass& dp = ctx->det_pass[pass];
  const int MB = ctx->cfg.max_batch, A = net.n_anchors, P = ctx->cfg.patch_size, N = dp.N;
  const int slot = det_slot_base(ctx) + pass;
  int rc;
  for (const Op& op : net.ops) {
    if (op.kind != OP_PRED) continue;
    float* g_reg = net.gact + net.buf_off[op.in.buf] * (size_t)MB + op.in.coff;
    float* g_cls = net.gact + net.buf_off[op.res.buf] * (size_t)MB + op.res.coff;
    rc = launch_head_pred_bwd(dp.dlogits, scale_dev, view_ptr(net, slot, MB, op.in), net.bufs[op.in.buf].C,
                              view_tab(net, slot, op.in), view_ptr(net, slot, MB, op.res), net.bufs[op.res.buf].C, view_tab(net, slot, op.res),
                              net.act_dtype, net.pred_w[op.level], g_reg, g_cls, grad_of(ctx, net.pred_w[op.level]),
                              grad_of(ctx, net.pred_b[op.level]), net.head_hid, op.in.H * op.in.W, A, op.anchor0, N, s);
    JN_CHECK(rc == 0, JN_ESTATE, "predictor backward: unsupported buffer type");
  }
  StemSrc ss{dp.patches, nullptr, 3LL * P * P, (long long)P * P, P};
  if ((rc = run_net_backward(ctx, JN_NET_DETECTOR, N, ss, slot, s, 1, 0, true))) return rc;
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_detector_step(jn_ctx* ctx, const float* patches_dev, int N, const float* targets_dev, int nb, float loss_scale,
                     float* metrics_dev, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = detector_forward_impl(ctx, patches_dev, N, targets_dev, nb, 0, 1, loss_scale, metrics_dev, s);
  if (rc) return rc;
  rc = detector_backward_impl(ctx, 0, ctx->det_pass[0].acc + 8, s);
  ctx->det_pass[0].valid = false;          // the gradient buffers of the head were consumed
  return rc;
}

// The eval-mode head on the TRAIN-mode FPN maps of a pass (src/models/yolox.py:74-91: `self.eval(); outputs =
// self.head(fpn_outs)` after the loss branch — fpn_outs were computed in the module's current mode, the head now uses
// its running statistics as the train pass has just updated them): the three FPN views (raw z + their batch-statistics
// table entries) are copied into the eval workspace (slot 0) and the head ops run there, so the pass's own head
// activations stay intact for the backward.
static int detector_eval_head(jn_ctx* ctx, int pass, float* boxes_dev, int32_t* counts_dev, hipStream_t s) {
  Net& net = ctx->nets[JN_NET_DETECTOR];
  const jn_ctx::DetPass& dp = ctx->det_pass[pass];
  const int MB = ctx->cfg.max_batch, N = dp.N, slot = det_slot_base(ctx) + pass;
  JN_CHECK(net.n_backbone_ops >= 0, JN_ESTATE, "detector without a head");
  int rc;
  if (!ctx->det_raw)
    if ((rc = dev_alloc(ctx, &ctx->det_raw, (size_t)MB * net.n_anchors * 6))) return rc;
  if ((rc = refresh_eval_table(ctx, net, s))) return rc;          // head tables from the running statistics (just updated)
  for (int i = 0; i < 3; ++i) {
    const View& f = net.fpn[i];
    const int ld = net.bufs[f.buf].C;
    launch_grad_copy((const float*)view_ptr(net, slot, MB, f), ld, (float*)view_ptr(net, 0, MB, f), ld, f.C, (long long)N * f.H * f.W, 0, s);
    const ChanTab src = view_tab(net, slot, f), dst = view_tab(net, 0, f);
    JN_HIP(hipMemcpyAsync(dst.sc, src.sc, f.C * sizeof(float), hipMemcpyDeviceToDevice, s));
    JN_HIP(hipMemcpyAsync(dst.sh, src.sh, f.C * sizeof(float), hipMemcpyDeviceToDevice, s));
    JN_HIP(hipMemcpyAsync(dst.fl, src.fl, f.C * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  StemSrc none{nullptr, nullptr, 0, 0, 0};
  rc = run_net(ctx, JN_NET_DETECTOR, N, none, 0, 0, nullptr, 0, s, true, net.n_backbone_ops);
  net.eval_tab_dirty = true;               // slot 0's FPN entries hold batch statistics now: rebuilt before the next eval pass
  if (rc) return rc;
  launch_postprocess(ctx->det_raw, net.n_anchors, N, ctx->cfg.det_conf_threshold, ctx->cfg.det_nms_threshold,
                     (float)(ctx->cfg.patch_size - 1), boxes_dev, counts_dev, ctx->cfg.max_det_per_patch, s);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_detector_forward(jn_ctx* ctx, const float* patches_dev, int N, const float* targets_dev, int nb, int pass, int n_pass,
                        float* metrics_dev, float* boxes_dev, int32_t* counts_dev, float* fpn0_dev, float* fpn1_dev,
                        float* fpn2_dev, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = detector_forward_impl(ctx, patches_dev, N, targets_dev, nb, pass, n_pass, 1.0f, metrics_dev, s);
  if (rc) return rc;
  Net& net = ctx->nets[JN_NET_DETECTOR];
  const int MB = ctx->cfg.max_batch, slot = det_slot_base(ctx) + pass;
  float* outs[3] = {fpn0_dev, fpn1_dev, fpn2_dev};
  for (int i = 0; i < 3; ++i) {
    if (!outs[i]) continue;
    const View& f = net.fpn[i];
    launch_nhwc_to_nchw(view_ptr(net, slot, MB, f), net.act_dtype, net.bufs[f.buf].C, view_tab(net, slot, f), outs[i], f.C, f.H * f.W, N, s);
  }
  if (boxes_dev && counts_dev && (rc = detector_eval_head(ctx, pass, boxes_dev, counts_dev, s))) return rc;
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_detector_backward(jn_ctx* ctx, int pass, const float* dloss_dev, float scale, void* stream) {
  JN_CHECK(ctx, JN_EINVAL, "jn_detector_backward: null ctx");
  JN_CHECK(ctx->has_net[JN_NET_DETECTOR], JN_ESTATE, "context was created without a detector");
  JN_CHECK(pass >= 0 && pass < (int)ctx->det_pass.size() && ctx->det_pass[pass].valid, JN_ESTATE,
           "jn_detector_backward: pass %d has no forward to differentiate (none ran, or a later pass overwrote its activations)", pass);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(det_scale_kernel, dim3(1), dim3(1), 0, s, ctx->det_pass[pass].acc + 8, dloss_dev, scale, ctx->det_bwd_scale);
  int rc = detector_backward_impl(ctx, pass, ctx->det_bwd_scale, s);
  ctx->det_pass[pass].valid = false;       // one backward per forward (retain_graph is not offered)
  return rc;
}

// ---- environment ---------------------------------------------------------------------
static EnvPtrs env_ptrs(jn_ctx* ctx) {
  const EnvState& e = ctx->env;
  EnvPtrs p{e.positions, e.bbox_masks, e.visited, e.steps, e.has_stopped, e.n_bbox_tiles, ctx->found,
            e.B, e.Gh, e.Gw, e.T, e.stop};
  return p;
}

int jn_env_init(jn_ctx* ctx, const float* images_dev, const int64_t* bboxes_dev, int B, int H, int W, int nb,
                int max_ep_len, int stop_enabled, void* stream) {
  JN_CHECK(ctx && images_dev && (bboxes_dev || nb == 0), JN_EINVAL, "jn_env_init: null argument");
  const int P = ctx->cfg.patch_size;
  JN_CHECK(B >= 1 && B <= ctx->cfg.max_batch, JN_EINVAL, "B=%d exceeds max_batch=%d", B, ctx->cfg.max_batch);
  // general_env.py:50-51
  JN_CHECK(H % P == 0 && W % P == 0, JN_EINVAL, "image %dx%d is not divisible by patch_size %d", H, W, P);
  JN_CHECK(H / P <= 256 && W / P <= 256, JN_EINVAL, "patch grid larger than 256");
  JN_CHECK(max_ep_len >= 1 && max_ep_len <= ctx->cfg.block_size, JN_EINVAL, "max_ep_len %d > block_size %d", max_ep_len,
           ctx->cfg.block_size);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  EnvState& e = ctx->env;
  const int Gh = H / P, Gw = W / P;
  int rc;
  if (!e.positions) {
    const int MB = ctx->cfg.max_batch;
    if ((rc = dev_alloc(ctx, &e.positions, (size_t)MB * 2))) return rc;
    if ((rc = dev_alloc(ctx, &e.bbox_masks, (size_t)MB * 256 * 256))) return rc;
    if ((rc = dev_alloc(ctx, &e.visited, (size_t)MB * 256 * 256))) return rc;
    if ((rc = dev_alloc(ctx, &e.steps, (size_t)MB))) return rc;
    if ((rc = dev_alloc(ctx, &e.has_stopped, (size_t)MB))) return rc;
    if ((rc = dev_alloc(ctx, &e.n_bbox_tiles, (size_t)MB))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->found, (size_t)MB))) return rc;
  }
  e.images = images_dev; e.B = B; e.H = H; e.W = W; e.nb = nb; e.Gh = Gh; e.Gw = Gw; e.T = max_ep_len;
  e.stop = stop_enabled ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  launch_bbox_masks(bboxes_dev, e.bbox_masks, e.n_bbox_tiles, B, nb, H, W, P, s);
  launch_env_reset(env_ptrs(ctx), nullptr, 0, s);   // zeroed state at (0,0)-independent start; reset() follows
  JN_HIP(hipGetLastError());
  e.ready = true;
  return JN_OK;
}

int jn_env_reset(jn_ctx* ctx, const int64_t* positions_dev, uint64_t seed, void* stream) {
  JN_CHECK(ctx && ctx->env.ready, JN_ESTATE, "jn_env_init has not been called");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  launch_env_reset(env_ptrs(ctx), positions_dev, seed, (hipStream_t)stream);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_env_step(jn_ctx* ctx, const int64_t* actions_dev, float* rewards_dev, uint8_t* terminated_dev,
                uint8_t* truncated_dev, void* stream) {
  JN_CHECK(ctx && ctx->env.ready, JN_ESTATE, "jn_env_init has not been called");
  JN_CHECK(actions_dev, JN_EINVAL, "jn_env_step: null actions");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  launch_env_step(env_ptrs(ctx), actions_dev, rewards_dev, terminated_dev, truncated_dev, (hipStream_t)stream);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_env_state(jn_ctx* ctx, int what, void** ptr_dev) {
  JN_CHECK(ctx && ptr_dev && ctx->env.ready, JN_ESTATE, "jn_env_init has not been called");
  switch (what) {
    case 0: *ptr_dev = ctx->env.positions; break;
    case 1: *ptr_dev = ctx->env.bbox_masks; break;
    case 2: *ptr_dev = ctx->env.visited; break;
    case 3: *ptr_dev = ctx->env.steps; break;
    case 4: *ptr_dev = ctx->env.has_stopped; break;
    default: set_error("jn_env_state: unknown selector %d", what); return JN_EINVAL;
  }
  return JN_OK;
}

int jn_gather_patches(const float* images_dev, const int64_t* positions_dev, float* out_dev, int B, int C, int H, int W,
                      int P, void* stream) {
  JN_CHECK(images_dev && positions_dev && out_dev, JN_EINVAL, "jn_gather_patches: null argument");
  JN_CHECK(B >= 0 && C >= 1 && P >= 1 && H % P == 0 && W % P == 0, JN_EINVAL, "jn_gather_patches: bad shape");
  if (B == 0) return JN_OK;
  launch_gather(images_dev, positions_dev, out_dev, (long long)C * P * P, B, C, H, W, P, nullptr, 0, (hipStream_t)stream);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_augment_patches(const float* in_dev, float* out_dev, const float* params_dev, const float* noise_dev, uint64_t seed,
                       int N, int P, void* stream) {
  JN_CHECK(in_dev && out_dev && params_dev, JN_EINVAL, "jn_augment_patches: null argument");
  JN_CHECK(in_dev != out_dev, JN_EINVAL, "jn_augment_patches: in-place is not supported (tiles read their neighbours' halo)");
  JN_CHECK(N >= 0 && P >= 4, JN_EINVAL, "jn_augment_patches: bad shape");
  if (N == 0) return JN_OK;
  launch_augment(in_dev, out_dev, params_dev, noise_dev, (unsigned long long)seed, N, P, (hipStream_t)stream);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_gather_patches_indexed(const float* images_dev, const int64_t* image_index_dev, const int64_t* positions_dev,
                              float* out_dev, int N, int n_images, int C, int H, int W, int P, void* stream) {
  JN_CHECK(images_dev && image_index_dev && positions_dev && out_dev, JN_EINVAL, "jn_gather_patches_indexed: null argument");
  JN_CHECK(N >= 0 && n_images >= 1 && C >= 1 && P >= 1 && H % P == 0 && W % P == 0, JN_EINVAL,
           "jn_gather_patches_indexed: bad shape");
  if (N == 0) return JN_OK;
  launch_gather(images_dev, positions_dev, out_dev, (long long)C * P * P, N, C, H, W, P, nullptr, 0, (hipStream_t)stream,
                image_index_dev);
  JN_HIP(hipGetLastError());
  return JN_OK;
}

int jn_env_patches(jn_ctx* ctx, float* out_dev, void* stream) {
  JN_CHECK(ctx && ctx->env.ready && out_dev, JN_ESTATE, "jn_env_init has not been called");
  const EnvState& e = ctx->env;
  return jn_gather_patches(e.images, e.positions, out_dev, e.B, 3, e.H, e.W, ctx->cfg.patch_size, stream);
}

// ---- the hot loop ----------------------------------------------------------------------
int jn_set_profiling(jn_ctx* ctx, int enabled) {
  JN_CHECK(ctx, JN_EINVAL, "null ctx");
  ctx->profiling = enabled != 0;
  return JN_OK;
}

static int rollout_impl(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev, const int64_t* start_positions_dev,
                        uint64_t seed, int do_detection, int stop_early, const jn_rollout_out* out, int train,
                        void* stream) {
  JN_CHECK(ctx && out, JN_EINVAL, "jn_rollout: null argument");
  JN_CHECK(ctx->env.ready, JN_ESTATE, "jn_env_init has not been called");
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(mode >= 0 && mode <= 2, JN_EINVAL, "unknown mode %d", mode);
  JN_CHECK(mode != JN_MODE_FORCED || forced_actions_dev, JN_EINVAL, "JN_MODE_FORCED needs forced_actions");
  JN_CHECK(out->rewards_dev && out->masks_dev, JN_EINVAL, "rewards_dev and masks_dev are required outputs");
  JN_CHECK(!do_detection || (ctx->has_net[JN_NET_DETECTOR] && out->det_boxes_dev && out->det_counts_dev), JN_EINVAL,
           "do_detection needs a detector and det_boxes_dev / det_counts_dev outputs");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  jn_ctx& x = *ctx;
  const jn_config& c = ctx->cfg;
  const EnvState& e = ctx->env;
  const int B = e.B, T = e.T, C = c.n_embd, P = c.patch_size, nA = c.n_actions;
  ctx->train_out_valid = false;       // every rollout restarts n_done / the env state the REINFORCE backward reads
  if (train) ctx->sup_valid = false;  // ... and a train-mode one the per-token buffers the supervised backward reads
  RolloutBuffers r{out->rewards_dev, out->returns_dev, out->logprobs_dev, out->entropies_dev, out->masks_dev,
                   out->logit_masks_dev, out->positions_dev, out->actions_dev, out->logits_dev, out->final_emb_dev};
  // zero-fill so that columns past an early stop read as the reference's absent columns would be cut
  JN_HIP(hipMemsetAsync(r.rewards, 0, (size_t)B * T * sizeof(float), s));
  if (r.returns) JN_HIP(hipMemsetAsync(r.returns, 0, (size_t)B * T * sizeof(float), s));
  if (r.logprobs) JN_HIP(hipMemsetAsync(r.logprobs, 0, (size_t)B * T * sizeof(float), s));
  if (r.entropies) JN_HIP(hipMemsetAsync(r.entropies, 0, (size_t)B * T * sizeof(float), s));
  JN_HIP(hipMemsetAsync(r.masks, 0, (size_t)B * (T + 1), s));
  if (r.logit_masks) JN_HIP(hipMemsetAsync(r.logit_masks, 0, (size_t)B * T, s));
  if (r.positions) JN_HIP(hipMemsetAsync(r.positions, 0, (size_t)B * (T + 1) * 2 * sizeof(int64_t), s));
  if (r.actions) JN_HIP(hipMemsetAsync(r.actions, 0, (size_t)B * T * sizeof(int64_t), s));
  if (r.logits) JN_HIP(hipMemsetAsync(r.logits, 0, (size_t)B * T * nA * sizeof(float), s));
  if (r.final_emb) JN_HIP(hipMemsetAsync(r.final_emb, 0, (size_t)B * (T + 1) * C * sizeof(float), s));

  if (ctx->ev[0]) JN_HIP(hipEventRecord(ctx->ev[0], s));
  EnvPtrs ep = env_ptrs(ctx);
  launch_env_reset(ep, start_positions_dev, seed, s);
  launch_rollout_begin(ep, r, ctx->prev_action, ctx->cache_len, ctx->n_done, s);
  const long long patch_stride = (long long)(T + 1) * 3 * P * P;
  if (out->patches_dev)
    launch_gather(e.images, e.positions, out->patches_dev, patch_stride, B, 3, e.H, e.W, P, nullptr, 0, s);

  const int Kd = c.max_det_per_patch;
  if (do_detection) {
    // src/reinforce.py:141-146 (start patch) — written for every image, see DESIGN.md deviations
    JN_HIP(hipMemsetAsync(out->det_counts_dev, 0, (size_t)B * (T + 1) * sizeof(int32_t), s));
    if (!ctx->det_tmp_boxes) {
      int rc2;
      if ((rc2 = dev_alloc(ctx, &ctx->det_tmp_boxes, (size_t)c.max_batch * Kd * 7))) return rc2;
      if ((rc2 = dev_alloc(ctx, &ctx->det_tmp_counts, (size_t)c.max_batch))) return rc2;
    }
  }
  // The detector pass of a glimpse only feeds the detection outputs, so it runs on the context's second stream beside
  // the next glimpse step of the decision path: the positions it reads are snapshotted per step (the agents move on),
  // forked after the step that produced them, joined before the rollout returns.
  static const bool no_aux_det = std::getenv("JN_NO_AUX_STREAM") != nullptr;
  hipStream_t ds_stream = s;
  // (only with a separate patch encoder: when the detector's own PAFPN encodes the patches — no gpt_backbone, the
  // reference's default — both passes use the slot-0 workspace and table of the same net and must stay in stream order)
  if (do_detection && !no_aux_det && ctx->enc_net != JN_NET_DETECTOR) {
    int ra = ensure_aux_stream(ctx);
    if (ra) return ra;
    if (!ctx->det_pos || ctx->det_pos_cap < (size_t)(T + 1) * B * 2) {
      if ((ra = dev_alloc(ctx, &ctx->det_pos, (size_t)(T + 1) * B * 2))) return ra;
      ctx->det_pos_cap = (size_t)(T + 1) * B * 2;
    }
    ds_stream = ctx->aux_stream;
  }
  auto detect_step = [&](int col, const int* flag) -> int {
    const int64_t* pos = e.positions;
    if (ds_stream != s) {
      int64_t* snap = ctx->det_pos + (size_t)col * B * 2;
      JN_HIP(hipMemcpyAsync(snap, e.positions, (size_t)B * 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
      JN_HIP(hipEventRecord(ctx->aux_fork, s));
      JN_HIP(hipStreamWaitEvent(ds_stream, ctx->aux_fork, 0));
      pos = snap;
    }
    StemSrc ds{e.images, pos, 3LL * e.H * e.W, (long long)e.H * e.W, e.W};
    int r = detect_impl(ctx, ds, B, ctx->det_tmp_boxes, ctx->det_tmp_counts, nullptr, flag, B, ds_stream);
    if (r) return r;
    launch_det_scatter(ctx->det_tmp_boxes, ctx->det_tmp_counts, out->det_boxes_dev, out->det_counts_dev, B, T + 1, col, Kd,
                       flag, B, ds_stream);
    return JN_OK;
  };
  if (do_detection) { int r0 = detect_step(0, nullptr); if (r0) return r0; }
  if (ctx->profiling) {
    while ((int)ctx->conv_ev.size() < 2 * T) {
      hipEvent_t ev;
      JN_HIP(hipEventCreate(&ev));
      ctx->conv_ev.push_back(ev);
    }
  }
  ctx->conv_ev_used = 0;
  StemSrc ss{e.images, e.positions, 3LL * e.H * e.W, (long long)e.H * e.W, e.W};
  int rc;
  if (train) {
    Net& tn = ctx->nets[ctx->enc_net];
    // (a shared detector / encoder net: the detector's training pass gets its slot behind the rollout's right away, so
    // that no growth — and copy — happens between this rollout and its backward)
    if ((rc = ensure_slots(ctx, tn, ctx->enc_net == JN_NET_DETECTOR ? c.block_size + 2 : T + 1))) return rc;
    int g_want = tn.g_slots;
    if (ctx->enc_net == JN_NET_DETECTOR) g_want = std::max(1, g_want);   // detached encoder: no conv-stack backward
    else if (g_want < T) {
      // gradient buffers for as many glimpse steps as fit in ~80 % of the free HBM (the whole trajectory on a
      // 288 GB MI355X at the headline sizes); the backward then runs in ceil(S / g_slots) step-batched passes
      size_t free_b = 0, total_b = 0;
      JN_HIP(hipMemGetInfo(&free_b, &total_b));
      const size_t per_slot = (size_t)tn.per_image_floats * c.max_batch * sizeof(float) + (1u << 20);
      const long long fit = (long long)((double)free_b * 0.8 / (double)per_slot);
      g_want = (int)std::max<long long>(1, std::min<long long>(T, fit));
      if (const char* cap = std::getenv("JN_GRAD_SLOTS")) g_want = std::max(1, std::min(g_want, std::atoi(cap)));
    }
    if ((rc = ensure_train_state(ctx, g_want))) return rc;
    if (!ctx->efpn_train) {
      const size_t MBt = (size_t)c.max_batch * c.block_size;
      if ((rc = dev_alloc(ctx, &ctx->efpn_train, MBt * ctx->efpn_h * ctx->efpn_w * C))) return rc;
      if ((rc = dev_alloc(ctx, &ctx->tok_emb_train, MBt * C))) return rc;
      if ((rc = dev_alloc(ctx, &ctx->d_tok_emb, MBt * C))) return rc;
      if ((rc = dev_alloc(ctx, &ctx->dlogits, MBt * nA))) return rc;
      if ((rc = dev_alloc(ctx, &ctx->de_ws, MBt * ctx->efpn_h * ctx->efpn_w * C))) return rc;
    }
  }
  if (train) ctx->drop_seed_used = ctx->drop_seed + ctx->drop_ctr++;   // dropout masks of this trajectory (regenerated in the backward)
  struct PrezeroGuard { jn_ctx* c; ~PrezeroGuard() { c->stats_prezeroed = false; } } prezero_guard{ctx};
  if (train && !c.no_patch_emb) {
    Net& tn = ctx->nets[ctx->enc_net];
    JN_HIP(hipMemsetAsync(slot_stats(tn, 1), 0, (size_t)T * JN_NREP * 2 * tn.stat_channels * sizeof(double), s));
    ctx->stats_prezeroed = true;
  }
  for (int t = 0; t < T; ++t) {
    const int* flag = stop_early ? ctx->n_done + t : nullptr;
    if (!c.no_patch_emb) {
      if (ctx->profiling) JN_HIP(hipEventRecord(ctx->conv_ev[2 * t], s));
      if ((rc = run_net(ctx, ctx->enc_net, B, ss, train ? t + 1 : 0, train, flag, B, s))) return rc;
      if (ctx->profiling) { JN_HIP(hipEventRecord(ctx->conv_ev[2 * t + 1], s)); ctx->conv_ev_used = 2 * (t + 1); }
      if ((rc = run_embed_fpn(ctx, B, train ? t + 1 : 0, train ? ctx->efpn_train + (size_t)t * B * ctx->efpn_h * ctx->efpn_w * C : nullptr, flag, B, s))) return rc;
    }
    GptStepArgs a{};
    a.C = C; a.n_head = c.n_head; a.n_layer = c.n_layer; a.nA = nA; a.Tmax = c.block_size + 1; a.B = B; a.T = T;
    a.use_pos_emb = c.use_pos_emb; a.no_patch_emb = c.no_patch_emb; a.concat_emb = c.concat_emb;
    a.dec_pos_enc = c.decoder_pos_encoding; a.n_parts = n_parts(c);
    a.pe2_ch = (int)std::ceil(C / 4.0) * 2;
    const GptW& g = ctx->gpt;
    a.wte = g.wte; a.wpe = g.wpe; a.embed_class = g.embed_class; a.proj_wt = g.proj_wt; a.proj_b = g.proj_b;
    a.pos1d = g.pos1d; a.pe2 = g.pos2d_col; a.head_wt = g.head_wt; a.lnf_w = g.lnf_w; a.lnf_b = g.lnf_b;
    a.layers = x.layers_dev; a.emb_part = x.emb_part; a.KS = x.KS; a.efpn_lin_b = g.efpn_lin_b;
    a.kcache = ctx->kcache; a.vcache = ctx->vcache; a.prev_action = ctx->prev_action; a.cache_len = ctx->cache_len;
    a.step = t; a.mode = mode; a.forced = forced_actions_dev; a.seed = seed;
    a.src_mode = GPT_SRC_ENV; a.pos_index = 0; a.emb_stride = T + 1;
    a.env = ep; a.out = r; a.n_done = ctx->n_done;
    a.skip_flag = flag; a.skip_when = B;
    a.tok_emb_out = train ? ctx->tok_emb_train : nullptr;
    a.pdrop = train ? ctx->pdrop : 0.0f; a.drop_seed = ctx->drop_seed_used;
    launch_gpt_step(a, s);
    if (out->patches_dev)
      launch_gather(e.images, e.positions, out->patches_dev + (long long)(t + 1) * 3 * P * P, patch_stride, B, 3, e.H, e.W,
                    P, flag, B, s);
    if (do_detection && (rc = detect_step(t + 1, flag))) return rc;     // src/reinforce.py:162-167
  }
  if (ds_stream != s) {
    JN_HIP(hipEventRecord(ctx->aux_join, ds_stream));
    JN_HIP(hipStreamWaitEvent(s, ctx->aux_join, 0));
  }
  launch_rollout_epilogue(r, ctx->n_done, B, T, stop_early ? 1 : 0, s);
  ctx->last_stop_early = stop_early != 0;
  if (ctx->ev[1]) JN_HIP(hipEventRecord(ctx->ev[1], s));
  JN_HIP(hipGetLastError());
  ctx->last_T = T;
  return JN_OK;
}

int jn_rollout(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev, const int64_t* start_positions_dev,
               uint64_t seed, int do_detection, int stop_early, const jn_rollout_out* out, void* stream) {
  return rollout_impl(ctx, mode, forced_actions_dev, start_positions_dev, seed, do_detection, stop_early, out, 0, stream);
}

int jn_rollout_steps(jn_ctx* ctx, int* n_steps, void* stream);

// ---- REINFORCE iteration ------------------------------------------------------------------
int jn_arena_info(jn_ctx* ctx, size_t* total_numel, size_t* optim_gpt_numel) {
  JN_CHECK(ctx && ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  if (total_numel) *total_numel = ctx->arena_size;
  if (optim_gpt_numel) *optim_gpt_numel = ctx->gpt_arena_end;
  return JN_OK;
}

int jn_set_grad_arena(jn_ctx* ctx, float* grads_dev, size_t numel) {
  JN_CHECK(ctx && grads_dev && ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(numel >= ctx->arena_size, JN_EINVAL, "gradient arena needs %zu floats, got %zu", ctx->arena_size, numel);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  int rc = ensure_train_state(ctx);
  if (rc) return rc;
  ctx->grads = grads_dev;         // caller-owned (e.g. a torch tensor handed to RCCL all-reduce)
  ctx->g_layers_dev = nullptr;    // gradient pointer table must be rebuilt
  return JN_OK;
}

static int build_grad_layer_table(jn_ctx* ctx) {
  if (ctx->g_layers_dev) return JN_OK;
  const int nL = ctx->cfg.n_layer;
  std::vector<GptLayerPtrs> gl(nL);
  for (int l = 0; l < nL; ++l) {
    const GptW::Layer& L = ctx->gpt.layers[l];
    gl[l] = GptLayerPtrs{grad_of(ctx, L.ln1_w), grad_of(ctx, L.ln1_b), grad_of(ctx, L.qkv_wt), grad_of(ctx, L.qkv_b),
                         grad_of(ctx, L.proj_wt), grad_of(ctx, L.proj_b), grad_of(ctx, L.ln2_w), grad_of(ctx, L.ln2_b),
                         grad_of(ctx, L.fc_wt), grad_of(ctx, L.fc_b), grad_of(ctx, L.fc2_wt), grad_of(ctx, L.fc2_b)};
  }
  GptLayerPtrs* d = nullptr;
  int rc = dev_alloc(ctx, &d, (size_t)nL);
  if (rc) return rc;
  JN_HIP(hipMemcpy(d, gl.data(), gl.size() * sizeof(GptLayerPtrs), hipMemcpyHostToDevice));
  ctx->g_layers_dev = d;
  return JN_OK;
}

// GPT backward over a trajectory: scratch (sized for either kernel) + launch.  The batched kernels are the default;
// JN_GPT_BWD_V1=1 keeps the one-workgroup-per-agent kernel (diagnostic), which also takes the shapes the batched one refuses.
static int launch_gpt_bwd(jn_ctx* ctx, GptBwdArgs& ba, hipStream_t s) {
  const jn_config& c = ctx->cfg;
  const int L = ba.T + 1, nL = c.n_layer, nh = c.n_head, C = c.n_embd;
  const long long per_agent = (long long)(nL + 1) * L * C + (long long)nL * (11LL * L * C + (long long)nh * L * L) +
                              12LL * L * C + (long long)nh * L * L + 4LL * C + 64;
  const size_t need = std::max((size_t)per_agent * c.max_batch,
                               gpt_backward_batched_scratch(C, nh, nL, c.n_actions, c.max_batch, ba.T));
  if (!ctx->gpt_bwd_scratch || ctx->gpt_bwd_scratch_floats < need) {
    int rc = dev_alloc(ctx, &ctx->gpt_bwd_scratch, need);
    if (rc) return rc;
    ctx->gpt_bwd_scratch_floats = need;
  }
  ba.scratch = ctx->gpt_bwd_scratch; ba.scratch_per_agent = per_agent;
  static const bool v1 = std::getenv("JN_GPT_BWD_V1") != nullptr;
  if (!v1) {
    std::vector<GptLayerPtrs> W(nL), G(nL);
    for (int l = 0; l < nL; ++l) {
      const GptW::Layer& y = ctx->gpt.layers[l];
      W[l] = GptLayerPtrs{y.ln1_w, y.ln1_b, y.qkv_wt, y.qkv_b, y.proj_wt, y.proj_b, y.ln2_w, y.ln2_b, y.fc_wt, y.fc_b, y.fc2_wt, y.fc2_b};
      G[l] = GptLayerPtrs{grad_of(ctx, y.ln1_w), grad_of(ctx, y.ln1_b), grad_of(ctx, y.qkv_wt), grad_of(ctx, y.qkv_b),
                          grad_of(ctx, y.proj_wt), grad_of(ctx, y.proj_b), grad_of(ctx, y.ln2_w), grad_of(ctx, y.ln2_b),
                          grad_of(ctx, y.fc_wt), grad_of(ctx, y.fc_b), grad_of(ctx, y.fc2_wt), grad_of(ctx, y.fc2_b)};
    }
    if (launch_gpt_backward_batched(ba, W.data(), G.data(), s) == 0) return JN_OK;
  }
  launch_gpt_backward(ba, s);
  return JN_OK;
}

}  // extern "C"
static bool bf16_train_allowed() { static const bool on = std::getenv("JN_ALLOW_BF16_TRAIN") != nullptr; return on; }
static int reinforce_backward_impl(jn_ctx* ctx, const jn_rollout_out* out, int S, int stop_early, hipStream_t s);
static int check_train_outputs(jn_ctx* ctx, const jn_rollout_out* out) {
  JN_CHECK(out->logits_dev && out->actions_dev && out->returns_dev && out->logit_masks_dev && out->positions_dev &&
               out->final_emb_dev && out->rewards_dev && out->masks_dev,
           JN_EINVAL, "training needs logits/actions/returns/logit_masks/positions/final_emb/rewards/masks outputs");
  JN_CHECK(!ctx->cfg.no_patch_emb, JN_ESTATE, "training without a patch encoder is not supported");
  JN_CHECK(ctx->cfg.block_size <= 62, JN_EINVAL, "training supports block_size <= 62");
  // batch-statistics BatchNorm on bf16-rounded pre-activations is ill-conditioned (JN_ALLOW_BF16_TRAIN=1
  // lifts the refusal for the measurement behind that statement, tools/bf16_train_probe.py)
  JN_CHECK(ctx->cfg.act_dtype == JN_F32 || bf16_train_allowed(), JN_ESTATE, "training needs act_dtype = fp32 (bf16 is the inference mode)");
  return JN_OK;
}
extern "C" {

// Autograd bridge (SURVEY.md §8b "jn_rollout_backward"): the train-mode rollout alone ...
int jn_reinforce_forward(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev, const int64_t* start_positions_dev,
                         uint64_t seed, int stop_early, const jn_rollout_out* out, void* stream) {
  JN_CHECK(ctx && out, JN_EINVAL, "jn_reinforce_forward: null argument");
  int rc = check_train_outputs(ctx, out);
  if (rc) return rc;
  if ((rc = rollout_impl(ctx, mode, forced_actions_dev, start_positions_dev, seed, 0, stop_early, out, 1, stream))) return rc;
  ctx->train_out = *out; ctx->train_out_valid = true;
  return JN_OK;
}

// ... and its backward for GIVEN upstream gradients of the rollout's logprobs / entropies [B, T] (what torch autograd
// hands to the rollout node when the caller differentiates any loss built from them, src/reinforce.py:217-265, 341).
int jn_reinforce_backward(jn_ctx* ctx, const float* dlogprobs_dev, const float* dentropies_dev, void* stream) {
  JN_CHECK(ctx && (dlogprobs_dev || dentropies_dev), JN_EINVAL, "jn_reinforce_backward: null argument");
  JN_CHECK(ctx->train_out_valid, JN_ESTATE, "jn_reinforce_backward needs a preceding jn_reinforce_forward / jn_reinforce_step");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const jn_rollout_out* out = &ctx->train_out;
  const EnvState& e = ctx->env;
  int S = 0, rc;
  if ((rc = jn_rollout_steps(ctx, &S, stream))) return rc;
  launch_logits_grad(out->logits_dev, out->actions_dev, dlogprobs_dev, dentropies_dev, ctx->n_done, ctx->dlogits, e.B, e.T,
                     ctx->cfg.n_actions, ctx->last_stop_early ? 1 : 0, s);
  return reinforce_backward_impl(ctx, out, S, ctx->last_stop_early ? 1 : 0, s);
}

int jn_reinforce_step(jn_ctx* ctx, int mode, const int64_t* forced_actions_dev, const int64_t* start_positions_dev,
                      uint64_t seed, int stop_early, const jn_train_opts* opts, const jn_rollout_out* out,
                      float* metrics_dev, void* stream) {
  JN_CHECK(ctx && opts && out && metrics_dev, JN_EINVAL, "jn_reinforce_step: null argument");
  JN_CHECK(opts->struct_size == (int)sizeof(jn_train_opts), JN_EINVAL, "jn_train_opts.struct_size mismatch");
  { int rc0 = check_train_outputs(ctx, out); if (rc0) return rc0; }
  hipStream_t s = (hipStream_t)stream;
  int rc = rollout_impl(ctx, mode, forced_actions_dev, start_positions_dev, seed, 0, stop_early, out, 1, stream);
  if (rc) return rc;
  ctx->train_out = *out; ctx->train_out_valid = true;
  const EnvState& e = ctx->env;
  const int B = e.B, T = e.T, nA = ctx->cfg.n_actions;
  int S = 0;
  if ((rc = jn_rollout_steps(ctx, &S, stream))) return rc;      // the one host sync of the iteration

  LossArgs la{};
  la.logits = out->logits_dev; la.actions = out->actions_dev; la.returns = out->returns_dev; la.rewards = out->rewards_dev;
  la.logit_masks = out->logit_masks_dev; la.n_done = ctx->n_done; la.dlogits = ctx->dlogits; la.metrics = metrics_dev;
  la.B = B; la.T = T; la.nA = nA; la.stop_early = stop_early ? 1 : 0; la.reward_norm = opts->reward_norm;
  la.ret_mean = opts->ret_mean; la.ret_std = opts->ret_std; la.entropy_weight = opts->entropy_weight;
  la.scale = opts->loss_scale;
  launch_reinforce_loss(la, s);
  return reinforce_backward_impl(ctx, out, S, stop_early, s);
}

// loss.backward() of a train-mode rollout given d loss / d logits in ctx->dlogits: causal GPT over the trajectory
// (teacher-forced recompute), embed_fpn, then the patch encoder of all S executed glimpse steps, step-batched.
static int reinforce_backward_impl(jn_ctx* ctx, const jn_rollout_out* out, int S, int stop_early, hipStream_t s) {
  int rc;
  if ((rc = build_grad_layer_table(ctx))) return rc;
  const jn_config& c = ctx->cfg;
  const EnvState& e = ctx->env;
  const int B = e.B, T = e.T, C = c.n_embd, nA = c.n_actions, P = c.patch_size;
  const int nL = c.n_layer, nh = c.n_head;
  const GptW& g = ctx->gpt;
  GptBwdArgs ba{};
  ba.C = C; ba.n_head = nh; ba.n_layer = nL; ba.nA = nA; ba.B = B; ba.T = T; ba.stop_early = stop_early ? 1 : 0;
  ba.use_pos_emb = c.use_pos_emb; ba.no_patch_emb = c.no_patch_emb; ba.concat_emb = c.concat_emb;
  ba.dec_pos_enc = c.decoder_pos_encoding; ba.pe2_ch = (int)std::ceil(C / 4.0) * 2;
  ba.n_done = ctx->n_done; ba.final_emb = out->final_emb_dev; ba.dlogits = ctx->dlogits; ba.actions = out->actions_dev;
  ba.positions = out->positions_dev; ba.pos_tokens = T + 1; ba.pos1d_by_token = 0; ba.tok_actions = nullptr;
  ba.tok_emb = ctx->tok_emb_train; ba.d_tok_emb = ctx->d_tok_emb;
  ba.dte_stride_b = 1; ba.dte_stride_t = B;          // [T][B][C]: the rows of one glimpse step are contiguous
  ba.wte = g.wte; ba.wpe = g.wpe; ba.proj_wt = g.proj_wt; ba.pos1d = g.pos1d; ba.pe2 = g.pos2d_col; ba.head_wt = g.head_wt;
  ba.lnf_w = g.lnf_w; ba.lnf_b = g.lnf_b; ba.layers = ctx->layers_dev; ba.g_layers = ctx->g_layers_dev;
  ba.g_wte = grad_of(ctx, g.wte); ba.g_wpe = g.wpe ? grad_of(ctx, g.wpe) : nullptr;
  ba.g_embed_class = grad_of(ctx, g.embed_class);
  ba.g_proj_wt = g.proj_wt ? grad_of(ctx, g.proj_wt) : nullptr; ba.g_proj_b = g.proj_b ? grad_of(ctx, g.proj_b) : nullptr;
  ba.g_head_wt = grad_of(ctx, g.head_wt); ba.g_lnf_w = grad_of(ctx, g.lnf_w); ba.g_lnf_b = grad_of(ctx, g.lnf_b);
  ba.pdrop = ctx->pdrop; ba.drop_seed = ctx->drop_seed_used; ba.Tmax = c.block_size + 1;
  if ((rc = launch_gpt_bwd(ctx, ba, s))) return rc;

  // patch-encoder side: all executed glimpse steps in ONE set of launches (chunks of net.g_slots steps
  // when the gradient buffers of a whole trajectory do not fit): embed_fpn backward, then the PAFPN.
  Net& net = ctx->nets[ctx->enc_net];
  const int MB = c.max_batch, HW = ctx->efpn_h * ctx->efpn_w, K = HW * C;
  const View& f2 = net.fpn[2];
  const ChanTab ident{ctx->ident, ctx->ident + 2048, ctx->ident + 4096};
  const long long g_slot = (long long)net.per_image_floats * MB;
  if (ctx->profiling && ctx->ev[2]) JN_HIP(hipEventRecord(ctx->ev[2], s));     // conv-stack backward section (bench.py)
  // (a detached encoder needs no gradient slots: all steps in one chunk)
  const int chunk = ctx->enc_net == JN_NET_DETECTOR ? std::max(S, 1) : net.g_slots;
  for (int t0 = 0; t0 < S; t0 += chunk) {
    const int g_n = std::min(chunk, S - t0);
    const long long Mr = (long long)g_n * B;                       // (step, agent) rows of this chunk
    const float* e_c = ctx->efpn_train + (size_t)t0 * B * K;       // [g_n*B][K] embed_fpn.0 activations
    const float* dpe = ctx->d_tok_emb + (size_t)t0 * B * C;        // [g_n*B][C]  d loss / d patch embedding
    // Linear backward as two GEMMs on the 1x1-conv kernels: de = dpe . W^T (then the ReLU mask), dW = e^T . dpe
    ConvArgs la2{};
    la2.in = dpe; la2.in_ld = C; la2.in_dtype = JN_F32; la2.itab = ident; la2.w = g.efpn_lin_wt; la2.bias = nullptr;
    la2.out = ctx->de_ws; la2.out_ld = K; la2.out_dtype = JN_F32; la2.bf16_mfma = 0;
    la2.N = (int)Mr; la2.H = 1; la2.W = 1; la2.OH = 1; la2.OW = 1; la2.cin = C; la2.cout = K; la2.stride = 1; la2.act = ACT_NONE;
    launch_pw(la2, s);
    launch_relu_mask(ctx->de_ws, e_c, Mr * K, s);
    launch_pw_bwd_weight(e_c, K, dpe, JN_F32, C, ident, grad_of(ctx, g.efpn_lin_wt), nullptr, Mr, K, C, s);
    launch_colsum_add(dpe, Mr, C, grad_of(ctx, g.efpn_lin_b), s);
    // embed_fpn.0 (1x1 conv) backward into the f2 gradient view of every slot
    SlotBatch sb;
    sb.n = g_n; sb.act = g_slot; sb.grad = g_slot; sb.tab = 3LL * net.tab_channels;
    float* g_f2 = net.gact + net.buf_off[f2.buf] * (size_t)MB + f2.coff;
    ConvArgs a{};
    a.in = ctx->de_ws; a.in_ld = C; a.in_dtype = JN_F32; a.itab = ident; a.w = g.efpn_w; a.bias = nullptr;
    a.out = g_f2; a.out_ld = net.bufs[f2.buf].C; a.out_dtype = JN_F32; a.bf16_mfma = 0;
    a.N = B; a.H = f2.H; a.W = f2.W; a.OH = f2.H; a.OW = f2.W; a.cin = C; a.cout = f2.C; a.stride = 1; a.act = ACT_NONE;
    a.accumulate = 0; a.w_transposed = 1; a.in_identity = 1;
    a.n_slots = g_n; a.in_slot_stride = (long long)B * K; a.out_slot_stride = g_slot; a.tab_slot_stride = 0;
    // the detector's PAFPN as patch encoder is detached (src/models/gpt.py:376-380 "Do not backpropagate through
    // yolox"): the policy gradient stops at embed_fpn.0's weight
    const bool detached = ctx->enc_net == JN_NET_DETECTOR;
    if (!detached) launch_pw(a, s);
    launch_pw_bwd_weight(ctx->de_ws, C, view_ptr(net, t0 + 1, MB, f2), net.act_dtype, net.bufs[f2.buf].C, view_tab(net, t0 + 1, f2),
                         grad_of(ctx, g.efpn_w), ctx->wpart, (long long)B * HW, C, f2.C, s, sb, (long long)B * K);
    if (detached) continue;
    for (int i = 0; i < 2; ++i) {
      const View& f = net.fpn[i];
      for (int j = 0; j < g_n; ++j)
        JN_HIP(hipMemsetAsync(net.gact + (size_t)j * g_slot + net.buf_off[f.buf] * (size_t)MB + f.coff, 0,
                              (size_t)B * f.H * f.W * f.C * sizeof(float), s));
    }
    StemSrc ss{e.images, out->positions_dev + 2 * t0, 3LL * e.H * e.W, (long long)e.H * e.W, e.W};
    ss.pos_stride = 2 * (T + 1);
    // (only fpn[2] carries a gradient from outside the encoder: embed_fpn; the other two FPN views were zeroed above)
    if ((rc = run_net_backward(ctx, ctx->enc_net, B, ss, t0 + 1, s, g_n, 2, false, 0x3))) return rc;
  }
  if (ctx->profiling && ctx->ev[3]) { JN_HIP(hipEventRecord(ctx->ev[3], s)); ctx->bwd_timed = true; }
  (void)P;
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// Teacher-forced (supervised) step, src/supervised.py:863-902 with the detector term off, in two halves so that the
// reference's own loop can sit between them (GPT.forward -> its CE loss -> loss.backward(), the supervised autograd
// bridge): the forward runs GPT.forward on the full sequence (B*T patches through the encoder in ONE train-mode pass,
// 1-D positions 0..T-1) and leaves the logits in ctx->sup_logits; the backward takes d loss / d logits in ctx->dlogits.
static int supervised_forward_impl(jn_ctx* ctx, const float* patches_dev, const int64_t* current_actions_dev,
                                   const int64_t* classes_dev, const int64_t* positions_dev, int B, int T, float* logits_out_dev,
                                   float* final_emb_out_dev, hipStream_t s) {
  JN_CHECK(ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  const jn_config& c = ctx->cfg;
  JN_CHECK(!c.no_patch_emb, JN_ESTATE, "training without a patch encoder is not supported");
  JN_CHECK(c.act_dtype == JN_F32 || bf16_train_allowed(), JN_ESTATE, "training needs act_dtype = fp32 (bf16 is the inference mode)");
  JN_CHECK(T >= 1 && T <= c.block_size && c.block_size <= 62, JN_EINVAL, "sequence length %d out of range", T);
  JN_CHECK(B >= 1 && B * T <= c.max_batch, JN_EINVAL, "B*T = %d patches exceed max_batch = %d", B * T, c.max_batch);
  JN_CHECK(!c.use_pos_emb || positions_dev, JN_EINVAL, "positions are required when use_pos_emb is set");
  JN_HIP(hipSetDevice(c.device));
  int rc;
  ctx->train_out_valid = false;     // the per-token training buffers (efpn_train, tok_emb_train, dropout seed) are reused
  if ((rc = ensure_train_state(ctx))) return rc;
  if ((rc = build_grad_layer_table(ctx))) return rc;
  const int C = c.n_embd, nA = c.n_actions, P = c.patch_size, N = B * T, L = T + 1;
  const int HW = ctx->efpn_h * ctx->efpn_w;
  if (!ctx->efpn_train) {
    const size_t MBt = (size_t)c.max_batch * c.block_size;
    if ((rc = dev_alloc(ctx, &ctx->efpn_train, MBt * HW * C))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->tok_emb_train, MBt * C))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tok_emb, MBt * C))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dlogits, MBt * nA))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->de_ws, MBt * HW * C))) return rc;
  }
  if (!ctx->sup_final_emb) {
    if ((rc = dev_alloc(ctx, &ctx->sup_final_emb, (size_t)c.max_batch * (c.block_size + 1) * C))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->sup_logits, (size_t)c.max_batch * c.block_size * nA))) return rc;
  }
  // ---- forward: encoder over all B*T patches at once (BN statistics over B*T) ----
  StemSrc ss{patches_dev, nullptr, 3LL * P * P, (long long)P * P, P};
  if ((rc = run_net(ctx, ctx->enc_net, N, ss, 0, 1, nullptr, 0, s))) return rc;
  if ((rc = run_embed_fpn(ctx, N, 0, ctx->efpn_train, nullptr, 0, s))) return rc;
  hipLaunchKernelGGL(emb_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, ctx->emb_part, ctx->gpt.efpn_lin_b,
                     ctx->tok_emb_train, (long long)C, N, ctx->KS, C);
  JN_HIP(hipMemsetAsync(ctx->cache_len, 0, (size_t)B * sizeof(int32_t), s));
  ctx->drop_seed_used = ctx->drop_seed + ctx->drop_ctr++;
  GptStepArgs a{};
  fill_gpt_weights(ctx, a);
  a.pdrop = ctx->pdrop; a.drop_seed = ctx->drop_seed_used;
  a.classes = classes_dev;
  a.B = B; a.T = L; a.emb_stride = L; a.out.final_emb = ctx->sup_final_emb; a.logits_stride = T * nA;
  for (int tok = 0; tok < L; ++tok) {
    a.step = tok;
    a.logits_rows = tok >= 1 ? ctx->sup_logits + (size_t)(tok - 1) * nA : nullptr;
    if (tok == 0) {
      a.src_mode = GPT_SRC_CLASS;
    } else {
      a.src_mode = GPT_SRC_TEACH;
      a.t_actions = current_actions_dev; a.t_positions = positions_dev; a.t_stride = T; a.t_index = tok - 1;
      a.pos_index = tok - 1;
      a.tok_emb = ctx->tok_emb_train; a.tok_emb_stride = T; a.tok_emb_index = tok - 1;
    }
    launch_gpt_step(a, s);
  }
  if (logits_out_dev)
    JN_HIP(hipMemcpyAsync(logits_out_dev, ctx->sup_logits, (size_t)N * nA * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (final_emb_out_dev)
    JN_HIP(hipMemcpyAsync(final_emb_out_dev, ctx->sup_final_emb, (size_t)B * L * C * sizeof(float), hipMemcpyDeviceToDevice, s));
  ctx->sup = {patches_dev, current_actions_dev, positions_dev, classes_dev, B, T};
  ctx->sup_valid = true;      // (run_net on slot 0 of the encoder cleared it: set last)
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// backward of the forward above for d loss / d logits [B][T][nA] in ctx->dlogits
static int supervised_backward_impl(jn_ctx* ctx, hipStream_t s) {
  JN_CHECK(ctx->sup_valid, JN_ESTATE,
           "the activations of the supervised forward were overwritten (another pass used the encoder's workspace) or no forward ran");
  const jn_config& c = ctx->cfg;
  const int B = ctx->sup.B, T = ctx->sup.T;
  const int64_t* current_actions_dev = ctx->sup.actions; const int64_t* positions_dev = ctx->sup.positions;
  const int C = c.n_embd, nA = c.n_actions, P = c.patch_size, N = B * T;
  const int HW = ctx->efpn_h * ctx->efpn_w, K = HW * C;
  int rc;
  Net& net = ctx->nets[ctx->enc_net];
  StemSrc ss{ctx->sup.patches, nullptr, 3LL * P * P, (long long)P * P, P};
  const int nL = c.n_layer, nh = c.n_head;
  const GptW& g = ctx->gpt;
  GptBwdArgs ba{};
  ba.C = C; ba.n_head = nh; ba.n_layer = nL; ba.nA = nA; ba.B = B; ba.T = T; ba.stop_early = 0;
  ba.use_pos_emb = c.use_pos_emb; ba.no_patch_emb = c.no_patch_emb; ba.concat_emb = c.concat_emb;
  ba.dec_pos_enc = c.decoder_pos_encoding; ba.pe2_ch = (int)std::ceil(C / 4.0) * 2;
  ba.n_done = ctx->n_done; ba.final_emb = ctx->sup_final_emb; ba.dlogits = ctx->dlogits; ba.actions = current_actions_dev;
  ba.tok_actions = current_actions_dev; ba.positions = positions_dev; ba.pos_tokens = T; ba.pos1d_by_token = 1;
  ba.classes = ctx->sup.classes;
  ba.tok_emb = ctx->tok_emb_train; ba.d_tok_emb = ctx->d_tok_emb; ba.dte_stride_b = T; ba.dte_stride_t = 1;
  ba.wte = g.wte; ba.wpe = g.wpe; ba.proj_wt = g.proj_wt; ba.pos1d = g.pos1d; ba.pe2 = g.pos2d_col; ba.head_wt = g.head_wt;
  ba.lnf_w = g.lnf_w; ba.lnf_b = g.lnf_b; ba.layers = ctx->layers_dev; ba.g_layers = ctx->g_layers_dev;
  ba.g_wte = grad_of(ctx, g.wte); ba.g_wpe = g.wpe ? grad_of(ctx, g.wpe) : nullptr;
  ba.g_embed_class = grad_of(ctx, g.embed_class);
  ba.g_proj_wt = g.proj_wt ? grad_of(ctx, g.proj_wt) : nullptr; ba.g_proj_b = g.proj_b ? grad_of(ctx, g.proj_b) : nullptr;
  ba.g_head_wt = grad_of(ctx, g.head_wt); ba.g_lnf_w = grad_of(ctx, g.lnf_w); ba.g_lnf_b = grad_of(ctx, g.lnf_b);
  ba.pdrop = ctx->pdrop; ba.drop_seed = ctx->drop_seed_used; ba.Tmax = c.block_size + 1;
  if ((rc = launch_gpt_bwd(ctx, ba, s))) return rc;
  const int MB = c.max_batch;
  const View& f2 = net.fpn[2];
  const ChanTab ident{ctx->ident, ctx->ident + 2048, ctx->ident + 4096};
  {   // Linear backward as two GEMMs (see jn_reinforce_step)
    ConvArgs la2{};
    la2.in = ctx->d_tok_emb; la2.in_ld = C; la2.in_dtype = JN_F32; la2.itab = ident; la2.w = g.efpn_lin_wt; la2.bias = nullptr;
    la2.out = ctx->de_ws; la2.out_ld = K; la2.out_dtype = JN_F32; la2.bf16_mfma = 0;
    la2.N = N; la2.H = 1; la2.W = 1; la2.OH = 1; la2.OW = 1; la2.cin = C; la2.cout = K; la2.stride = 1; la2.act = ACT_NONE;
    launch_pw(la2, s);
    launch_relu_mask(ctx->de_ws, ctx->efpn_train, (long long)N * K, s);
    launch_pw_bwd_weight(ctx->efpn_train, K, ctx->d_tok_emb, JN_F32, C, ident, grad_of(ctx, g.efpn_lin_wt), nullptr, N, K, C, s);
    launch_colsum_add(ctx->d_tok_emb, N, C, grad_of(ctx, g.efpn_lin_b), s);
  }
  float* g_f2 = net.gact + net.buf_off[f2.buf] * (size_t)MB + f2.coff;
  ConvArgs ca{};
  ca.in = ctx->de_ws; ca.in_ld = C; ca.in_dtype = JN_F32; ca.itab = ident; ca.w = g.efpn_w; ca.bias = nullptr;
  ca.out = g_f2; ca.out_ld = net.bufs[f2.buf].C; ca.out_dtype = JN_F32; ca.bf16_mfma = net.act_dtype == JN_BF16;
  ca.N = N; ca.H = f2.H; ca.W = f2.W; ca.OH = f2.H; ca.OW = f2.W; ca.cin = C; ca.cout = f2.C; ca.stride = 1; ca.act = ACT_NONE;
  ca.accumulate = 0; ca.w_transposed = 1; ca.in_identity = 1;
  const bool detached = ctx->enc_net == JN_NET_DETECTOR;        // src/models/gpt.py:376-380, see jn_reinforce_step
  if (!detached) launch_pw(ca, s);
  launch_pw_bwd_weight(ctx->de_ws, C, view_ptr(net, 0, MB, f2), net.act_dtype, net.bufs[f2.buf].C, view_tab(net, 0, f2),
                       grad_of(ctx, g.efpn_w), ctx->wpart, (long long)N * HW, C, f2.C, s);
  if (detached) { JN_HIP(hipGetLastError()); return JN_OK; }
  for (int i = 0; i < 2; ++i) {
    const View& f = net.fpn[i];
    JN_HIP(hipMemsetAsync(net.gact + net.buf_off[f.buf] * (size_t)MB + f.coff, 0, (size_t)N * f.H * f.W * f.C * sizeof(float), s));
  }
  if ((rc = run_net_backward(ctx, ctx->enc_net, N, ss, 0, s, 1, 0, false, 0x3))) return rc;
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// One supervised (teacher-forced) training step minus the optimiser: forward, CrossEntropy(weight[STOP] = stop_weight)
// over non-padding tokens, backward.
int jn_supervised_step(jn_ctx* ctx, const float* patches_dev, const int64_t* current_actions_dev,
                       const int64_t* next_actions_dev, const int64_t* classes_dev, const int64_t* positions_dev,
                       const uint8_t* masks_dev, int B, int T, float stop_weight, float* logits_out_dev, float* metrics_dev,
                       void* stream) {
  JN_CHECK(ctx && patches_dev && current_actions_dev && next_actions_dev && masks_dev && metrics_dev, JN_EINVAL,
           "jn_supervised_step: null argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = supervised_forward_impl(ctx, patches_dev, current_actions_dev, classes_dev, positions_dev, B, T, logits_out_dev, nullptr, s);
  if (rc) return rc;
  launch_ce_loss(ctx->sup_logits, next_actions_dev, masks_dev, stop_weight, ctx->dlogits, metrics_dev, B * T, ctx->cfg.n_actions, T, s);
  return supervised_backward_impl(ctx, s);
}

// Supervised autograd bridge: GPT.forward(patches [B,T,3,P,P], actions [B,T], classes = 0, positions [B,T,2]) in train
// mode (src/models/gpt.py:481-534 as called by src/supervised.py:863-868) -> logits [B,T,nA], final_emb [B,T+1,C].  The
// input buffers must stay alive until jn_supervised_backward.
int jn_supervised_forward(jn_ctx* ctx, const float* patches_dev, const int64_t* current_actions_dev, const int64_t* classes_dev,
                          const int64_t* positions_dev, int B, int T, float* logits_out_dev, float* final_emb_out_dev, void* stream) {
  JN_CHECK(ctx && patches_dev && current_actions_dev && logits_out_dev, JN_EINVAL, "jn_supervised_forward: null argument");
  return supervised_forward_impl(ctx, patches_dev, current_actions_dev, classes_dev, positions_dev, B, T, logits_out_dev,
                                 final_emb_out_dev, (hipStream_t)stream);
}

// ... and its backward for GIVEN d loss / d logits [B,T,nA] (what torch hands to the logits node when the caller's
// loss.backward() runs, src/supervised.py:897): parameter gradients accumulate in the gradient arena.
int jn_supervised_backward(jn_ctx* ctx, const float* dlogits_dev, void* stream) {
  JN_CHECK(ctx && dlogits_dev, JN_EINVAL, "jn_supervised_backward: null argument");
  JN_CHECK(ctx->sup_valid, JN_ESTATE,
           "jn_supervised_backward: no supervised forward to differentiate (none ran, or a later pass overwrote its activations)");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  JN_HIP(hipMemcpyAsync(ctx->dlogits, dlogits_dev, (size_t)ctx->sup.B * ctx->sup.T * ctx->cfg.n_actions * sizeof(float),
                        hipMemcpyDeviceToDevice, s));
  return supervised_backward_impl(ctx, s);
}

int jn_optimizer_step(jn_ctx* ctx, float lr, float weight_decay, float clip_value, float grad_scale, void* stream) {
  return jn_optimizer_step_group(ctx, 0, lr, weight_decay, clip_value, grad_scale, stream);
}

int jn_optimizer_step_group(jn_ctx* ctx, int group, float lr, float weight_decay, float clip_value, float grad_scale,
                            void* stream) {
  JN_CHECK(ctx && ctx->grads, JN_ESTATE, "no gradients: run jn_reinforce_step / jn_detector_step first");
  JN_CHECK(group == 0 || group == 1, JN_EINVAL, "parameter group %d: 0 = optim_gpt, 1 = optim_yolox", group);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  // a frozen detector backbone (requires_grad = False in the reference: torch's AdamW skips it) keeps its values; its
  // BatchNorm running statistics still move in train-mode passes, as torch's do
  const size_t lo = group == 0 ? 0 : (ctx->freeze_det_backbone ? std::max(ctx->gpt_arena_end, ctx->det_head_begin) : ctx->gpt_arena_end);
  const size_t hi = group == 0 ? ctx->gpt_arena_end : ctx->arena_used;
  JN_CHECK(hi > lo, JN_ESTATE, "parameter group %d is empty", group);
  int& step = group == 0 ? ctx->adam_step : ctx->adam_step_yolox;
  step += 1;
  launch_adamw(ctx->params + lo, ctx->grads + lo, ctx->adam_m + lo, ctx->adam_v + lo, (long long)(hi - lo), lr, 0.9f, 0.999f, 1e-8f,
               weight_decay, step, clip_value, grad_scale, (hipStream_t)stream);
  for (int ni = 0; ni < 2; ++ni) if (ctx->has_net[ni]) ctx->nets[ni].eval_tab_dirty = ctx->nets[ni].x3_dirty = true;   // BN affine, 1x1 weights moved
  JN_HIP(hipGetLastError());
  return JN_OK;
}

// ---- arena <-> reference layout on the device (autograd bridge: param.data / param.grad of the Python module are
// views of ONE reference-layout buffer whose segment offsets equal the arena's) ---------------------------------------
static int ensure_segs_dev(jn_ctx* ctx) {
  if (ctx->segs_dev && ctx->segs_dev_n == (int)ctx->segs.size()) return JN_OK;
  std::vector<ArenaSeg> h(ctx->segs.size());
  for (size_t i = 0; i < h.size(); ++i) {
    const ParamSeg& g = ctx->segs[i];
    h[i] = ArenaSeg{(long long)g.off, (long long)g.numel, g.kind, g.d0, g.d1, g.d2};
  }
  ArenaSeg* d = nullptr;
  int rc = dev_alloc(ctx, &d, h.size());
  if (rc) return rc;
  JN_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(ArenaSeg), hipMemcpyHostToDevice));
  ctx->segs_dev = d; ctx->segs_dev_n = (int)h.size();
  return JN_OK;
}

int jn_arena_segment(jn_ctx* ctx, const char* name, size_t* off, size_t* numel) {
  JN_CHECK(ctx && name && ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  auto it = ctx->seg_index.find(name);
  JN_CHECK(it != ctx->seg_index.end(), JN_ENOTFOUND, "jn_arena_segment: '%s' is not a trainable tensor", name);
  if (off) *off = ctx->segs[it->second].off;
  if (numel) *numel = ctx->segs[it->second].numel;
  return JN_OK;
}

static int arena_copy(jn_ctx* ctx, int what, float* ref_dev, size_t numel, int to_ref, int accumulate, void* stream) {
  JN_CHECK(ctx && ref_dev && ctx->weights_loaded, JN_ESTATE, "jn_load_weights has not been called");
  JN_CHECK(what >= 0 && what <= 3, JN_EINVAL, "what: 0 = parameters, 1 = gradients, 2 / 3 = AdamW first / second moments");
  JN_CHECK(numel >= ctx->arena_used, JN_EINVAL, "reference-layout buffer needs %zu floats, got %zu", ctx->arena_used, numel);
  JN_HIP(hipSetDevice(ctx->cfg.device));
  int rc;
  if (what >= 1 && (rc = ensure_train_state(ctx))) return rc;
  if ((rc = ensure_segs_dev(ctx))) return rc;
  float* arena = what == 0 ? ctx->params : what == 1 ? ctx->grads : what == 2 ? ctx->adam_m : ctx->adam_v;
  launch_arena_copy(ctx->segs_dev, ctx->segs_dev_n, arena, ref_dev, (long long)ctx->arena_used, to_ref, accumulate,
                    (hipStream_t)stream);
  JN_HIP(hipGetLastError());
  if (!to_ref && what == 0)
    for (int ni = 0; ni < 2; ++ni) if (ctx->has_net[ni]) ctx->nets[ni].eval_tab_dirty = ctx->nets[ni].x3_dirty = true;   // parameters may have moved
  return JN_OK;
}

int jn_export_arena(jn_ctx* ctx, int what, float* dst_dev, size_t numel, int accumulate, void* stream) {
  return arena_copy(ctx, what, dst_dev, numel, 1, accumulate, stream);
}

int jn_import_arena(jn_ctx* ctx, int what, const float* src_dev, size_t numel, void* stream) {
  return arena_copy(ctx, what, const_cast<float*>(src_dev), numel, 0, 0, stream);
}

int jn_set_freeze(jn_ctx* ctx, int freeze_detector_backbone) {
  JN_CHECK(ctx, JN_EINVAL, "null ctx");
  ctx->freeze_det_backbone = freeze_detector_backbone != 0;
  return JN_OK;
}

int jn_optimizer_steps(jn_ctx* ctx, int group, int* steps, int set) {
  JN_CHECK(ctx && steps && (group == 0 || group == 1), JN_EINVAL, "jn_optimizer_steps: bad argument");
  int& st = group == 0 ? ctx->adam_step : ctx->adam_step_yolox;
  if (set) st = *steps; else *steps = st;
  return JN_OK;
}

int jn_set_dropout(jn_ctx* ctx, float p, uint64_t seed) {
  JN_CHECK(ctx && p >= 0.0f && p < 1.0f, JN_EINVAL, "dropout probability must be in [0, 1)");
  ctx->pdrop = p; ctx->drop_seed = seed; ctx->drop_ctr = 0;
  return JN_OK;
}

int jn_read_param(jn_ctx* ctx, const char* name, float* host_out, size_t numel) {
  JN_CHECK(ctx && name && host_out && ctx->params, JN_EINVAL, "jn_read_param: bad argument");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  auto it = ctx->seg_index.find(name);
  JN_CHECK(it != ctx->seg_index.end(), JN_ENOTFOUND, "jn_read_param: '%s' is not a trainable tensor", name);
  const ParamSeg& sg = ctx->segs[it->second];
  JN_CHECK(sg.numel == numel, JN_EINVAL, "'%s' has %zu elements, not %zu", name, sg.numel, numel);
  std::vector<float> packed(numel);
  JN_HIP(hipDeviceSynchronize());
  JN_HIP(hipMemcpy(packed.data(), ctx->params + sg.off, numel * sizeof(float), hipMemcpyDeviceToHost));
  const std::vector<float> t = unpack_param(sg, packed);
  std::memcpy(host_out, t.data(), numel * sizeof(float));
  return JN_OK;
}

int jn_rollout_steps(jn_ctx* ctx, int* n_steps, void* stream) {
  JN_CHECK(ctx && n_steps && ctx->last_T > 0, JN_ESTATE, "no rollout has run");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  const int T = ctx->last_T, B = ctx->env.B;
  std::vector<int32_t> h(T + 1);
  JN_HIP(hipMemcpyAsync(h.data(), ctx->n_done, (T + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  JN_HIP(hipStreamSynchronize((hipStream_t)stream));
  int S = T;
  if (ctx->last_stop_early)
    for (int t = 1; t <= T; ++t)
      if (h[t] >= B) { S = t; break; }
  *n_steps = S;
  return JN_OK;
}

int jn_last_timing(jn_ctx* ctx, int what, float* ms) {
  JN_CHECK(ctx && ms && ctx->last_T > 0, JN_ESTATE, "no rollout has run");
  JN_HIP(hipSetDevice(ctx->cfg.device));
  JN_HIP(hipEventSynchronize(ctx->ev[1]));
  if (what == 0) {
    JN_HIP(hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
  } else if (what == 2) {
    JN_CHECK(ctx->bwd_timed, JN_ESTATE, "no profiled jn_reinforce_step has run");
    JN_HIP(hipEventSynchronize(ctx->ev[3]));
    JN_HIP(hipEventElapsedTime(ms, ctx->ev[2], ctx->ev[3]));
  } else {
    float tot = 0.0f;
    for (int i = 0; i + 1 < ctx->conv_ev_used; i += 2) {
      float m = 0.0f;
      JN_HIP(hipEventElapsedTime(&m, ctx->conv_ev[i], ctx->conv_ev[i + 1]));
      tot += m;
    }
    *ms = tot;
  }
  return JN_OK;
}

}  // extern "C"
