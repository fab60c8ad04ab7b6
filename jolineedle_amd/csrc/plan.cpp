// Layer plans: turns (depth, width, depthwise, P) into a flat list of conv ops over
// NHWC buffers, plus the state-dict table the weights are loaded by.
//
// Topology restated from the published YOLOX CSPDarknet / YOLOPAFPN / YOLOXHead
// (the package the reference imports at src/models/gpt.py:24, src/models/yolox.py:7-10;
// SURVEY.md §2.1).  Concats are free: producers write into channel slices of the
// consumer's buffer; nearest-x2 upsampling is a copy into a slice.
#include <cmath>
#include <cstring>

#include "jn_internal.h"

namespace jnr {

namespace {

struct Builder {
  Net& net;
  std::vector<ParamEntry>& params;
  std::string mod_prefix;         // state-dict prefix of the part being built ("gpt_backbone.", "yolox.head.")

  int new_buf(int H, int W, int C) {
    Buf b; b.H = H; b.W = W; b.C = C;
    net.bufs.push_back(b);
    return (int)net.bufs.size() - 1;
  }
  View full(int buf) const {
    const Buf& b = net.bufs[buf];
    View v; v.buf = buf; v.H = b.H; v.W = b.W; v.C = b.C; v.coff = 0;
    return v;
  }
  static View slice(View v, int coff, int C) { v.coff += coff; v.C = C; return v; }
  View fresh(int H, int W, int C) { return full(new_buf(H, W, C)); }

  void add_param(const std::string& name, std::initializer_list<int64_t> shape, int dtype, bool buffer, bool used) {
    ParamEntry e; std::memset(&e, 0, sizeof(e));
    std::snprintf(e.info.name, sizeof(e.info.name), "%s", name.c_str());
    e.info.dtype = dtype; e.info.ndim = (int)shape.size();
    int i = 0; for (auto s : shape) e.info.shape[i++] = s;
    e.info.is_buffer = buffer; e.info.used = used;
    params.push_back(e);
  }

  int add_conv(const std::string& name, int cin, int cout, int k, int groups, bool bn, bool bias) {
    ConvW w; w.prefix = mod_prefix + name; w.cin = cin; w.cout = cout; w.k = k; w.groups = groups;
    w.has_bn = bn; w.has_bias = bias;
    net.convs.push_back(w);
    if (bn) {
      add_param(w.prefix + ".conv.weight", {cout, cin / groups, k, k}, 0, false, true);
      add_param(w.prefix + ".bn.weight", {cout}, 0, false, true);
      add_param(w.prefix + ".bn.bias", {cout}, 0, false, true);
      add_param(w.prefix + ".bn.running_mean", {cout}, 0, true, true);
      add_param(w.prefix + ".bn.running_var", {cout}, 0, true, true);
      add_param(w.prefix + ".bn.num_batches_tracked", {}, 1, true, false);
    } else {
      add_param(w.prefix + ".weight", {cout, cin / groups, k, k}, 0, false, true);
      if (bias) add_param(w.prefix + ".bias", {cout}, 0, false, true);
    }
    return (int)net.convs.size() - 1;
  }

  // conv `a` (first rows) and conv `b` as one 1x1 conv of 2h output channels over the same input
  int add_conv_pair(const std::string& a, const std::string& b, int cin, int h) {
    ConvW w; w.prefix = mod_prefix + a; w.prefix2 = mod_prefix + b; w.cin = cin; w.cout = 2 * h; w.cout_first = h;
    w.k = 1; w.groups = 1; w.has_bn = true; w.has_bias = false;
    net.convs.push_back(w);
    for (const std::string& pre : {w.prefix, w.prefix2}) {
      add_param(pre + ".conv.weight", {h, cin, 1, 1}, 0, false, true);
      add_param(pre + ".bn.weight", {h}, 0, false, true);
      add_param(pre + ".bn.bias", {h}, 0, false, true);
      add_param(pre + ".bn.running_mean", {h}, 0, true, true);
      add_param(pre + ".bn.running_var", {h}, 0, true, true);
      add_param(pre + ".bn.num_batches_tracked", {}, 1, true, false);
    }
    return (int)net.convs.size() - 1;
  }

  static int out_dim(int x, int s) { return (x + 2 - 3) / s + 1; }   // 3x3, pad 1

  // BaseConv(cin, cout, k, s, groups): conv + BN + SiLU.
  View base_conv(const std::string& name, View in, int cout, int k, int s, bool dw,
                 const View* dst = nullptr, const View* res = nullptr) {
    int OH = (k == 1) ? in.H : out_dim(in.H, s), OW = (k == 1) ? in.W : out_dim(in.W, s);
    // with a shortcut the conv writes raw z to a scratch buffer and OP_ADDACT materialises z + res
    View out = (dst && !res) ? *dst : fresh(OH, OW, cout);
    Op op;
    op.kind = (k == 1) ? OP_PW : (dw ? OP_DW : OP_CONV3);
    op.in = in; op.out = out; op.stride = s; op.act = ACT_NONE; op.name = name;
    op.wslot = add_conv(name, in.C, cout, k, dw ? in.C : 1, true, false);
    net.ops.push_back(op);
    if (!res) return out;
    Op add; add.kind = OP_ADDACT; add.in = out; add.res = *res; add.act = ACT_NONE; add.name = name + "+shortcut";
    add.out = dst ? *dst : fresh(OH, OW, cout);
    net.ops.push_back(add);
    return add.out;
  }
  // DWConv = depthwise k x k (stride s) + pointwise 1x1.
  View dw_conv(const std::string& name, View in, int cout, int k, int s,
               const View* dst = nullptr, const View* res = nullptr) {
    View d = base_conv(name + ".dconv", in, in.C, k, s, true);
    return base_conv(name + ".pconv", d, cout, 1, 1, false, dst, res);
  }
  View conv(const std::string& name, View in, int cout, int k, int s,
            const View* dst = nullptr, const View* res = nullptr) {
    if (net.depthwise) return dw_conv(name, in, cout, k, s, dst, res);
    return base_conv(name, in, cout, k, s, false, dst, res);
  }
  // Bottleneck(c, c, shortcut, expansion=1.0)
  View bottleneck(const std::string& name, View in, bool shortcut, const View* dst) {
    View u = base_conv(name + ".conv1", in, in.C, 1, 1, false);
    return conv(name + ".conv2", u, in.C, 3, 1, dst, shortcut ? &in : nullptr);
  }
  // CSPLayer: conv1 and conv2 read the same input, so they run as ONE 1x1 conv of 2h channels writing the slices
  // [conv2 | conv1] of a 3h-channel buffer [m(conv1) | conv2 | conv1]; conv3 reads the first 2h channels
  // (= cat(m(conv1), conv2), the reference's order).  Halves the input reads and the launches of the pair, in
  // eval and train mode alike (BN statistics are per channel).
  View csp(const std::string& name, View in, int cout, int n, bool shortcut, const View* dst = nullptr) {
    int h = cout / 2;
    if (n == 0) {
      View cat = fresh(in.H, in.W, 2 * h);
      View s0 = slice(cat, 0, h), s1 = slice(cat, h, h);
      base_conv(name + ".conv1", in, h, 1, 1, false, &s0);
      base_conv(name + ".conv2", in, h, 1, 1, false, &s1);
      return base_conv(name + ".conv3", cat, cout, 1, 1, false, dst);
    }
    View buf3 = fresh(in.H, in.W, 3 * h);
    View s0 = slice(buf3, 0, h), pair = slice(buf3, h, 2 * h), t = slice(buf3, 2 * h, h);
    Op op;
    op.kind = OP_PW; op.in = in; op.out = pair; op.stride = 1; op.act = ACT_NONE; op.name = name + ".conv2|conv1";
    op.wslot = add_conv_pair(name + ".conv2", name + ".conv1", in.C, h);
    net.ops.push_back(op);
    for (int i = 0; i < n; ++i)
      t = bottleneck(name + ".m." + std::to_string(i), t, shortcut, i == n - 1 ? &s0 : nullptr);
    return base_conv(name + ".conv3", slice(buf3, 0, 2 * h), cout, 1, 1, false, dst);
  }
  View spp(const std::string& name, View in, int cout) {
    int h = in.C / 2;
    View cat = fresh(in.H, in.W, 4 * h);
    View s0 = slice(cat, 0, h);
    base_conv(name + ".conv1", in, h, 1, 1, false, &s0);
    Op op; op.kind = OP_SPP; op.in = s0; op.out = cat; op.name = name + ".m"; op.act = ACT_NONE;
    net.ops.push_back(op);
    return base_conv(name + ".conv2", cat, cout, 1, 1, false);
  }
  void upsample(View in, View dst) {
    // the upsampled copy holds raw z of the producing conv: it shares that layer's (scale, shift)
    for (auto it = net.ops.rbegin(); it != net.ops.rend(); ++it)
      if (it->wslot >= 0 && it->out.buf == in.buf && it->out.coff == in.coff) { it->alias = dst; break; }
    Op op; op.kind = OP_UPSAMPLE; op.in = in; op.out = dst; op.name = "upsample"; op.act = ACT_NONE;
    net.ops.push_back(op);
  }
};

}  // namespace

int build_pafpn(Net& net, std::vector<ParamEntry>& params, const std::string& prefix,
                float depth, float width, bool depthwise, int P) {
  JN_CHECK(P % 32 == 0 && P >= 32, JN_EINVAL, "patch_size %d must be a multiple of 32", P);
  net.prefix = prefix; net.depthwise = depthwise; net.depth = depth; net.width = width; net.P = P;
  Builder b{net, params, prefix};
  const int bc = (int)(width * 64);
  const int bd = std::max((int)std::lround(depth * 3), 1);
  const int c0 = (int)(256 * width), c1 = (int)(512 * width), c2 = (int)(1024 * width);
  const int n = (int)std::lround(3 * depth);
  JN_CHECK(bc % 16 == 0, JN_EINVAL, "width %.3f gives %d stem channels; channel counts must be multiples of 16", width, bc);
  JN_CHECK(c0 == bc * 4 && c1 == bc * 8 && c2 == bc * 16, JN_EINVAL, "unsupported width %.3f", width);
  const int H2 = P / 2, H4 = P / 4, H8 = P / 8, H16 = P / 16, H32 = P / 32;

  // concat buffers of the PAFPN, allocated first so producers can target their slices
  View cat_p4 = b.fresh(H16, H16, 2 * c1);
  View cat_p3 = b.fresh(H8, H8, 2 * c0);
  View cat_n3 = b.fresh(H16, H16, 2 * c0);
  View cat_n4 = b.fresh(H32, H32, 2 * c1);

  // ---- CSPDarknet ("backbone.backbone.*") ----
  View stem = b.fresh(H2, H2, bc);
  {
    Op op; op.kind = OP_STEM; op.out = stem; op.act = ACT_NONE; op.name = "backbone.stem.conv";
    op.in.buf = -1; op.in.H = P; op.in.W = P; op.in.C = 3;
    op.wslot = b.add_conv("backbone.stem.conv", 12, bc, 3, 1, true, false);
    net.ops.push_back(op);
  }
  View x = b.conv("backbone.dark2.0", stem, bc * 2, 3, 2);
  x = b.csp("backbone.dark2.1", x, bc * 2, bd, true);
  x = b.conv("backbone.dark3.0", x, bc * 4, 3, 2);
  View x2_dst = Builder::slice(cat_p3, c0, c0);
  View x2 = b.csp("backbone.dark3.1", x, bc * 4, bd * 3, true, &x2_dst);
  x = b.conv("backbone.dark4.0", x2, bc * 8, 3, 2);
  View x1_dst = Builder::slice(cat_p4, c1, c1);
  View x1 = b.csp("backbone.dark4.1", x, bc * 8, bd * 3, true, &x1_dst);
  x = b.conv("backbone.dark5.0", x1, bc * 16, 3, 2);
  x = b.spp("backbone.dark5.1", x, bc * 16);
  View x0 = b.csp("backbone.dark5.2", x, bc * 16, bd, false);
  (void)H4;

  // ---- PAFPN ----
  View fpn_out0_dst = Builder::slice(cat_n4, c1, c1);
  View fpn_out0 = b.base_conv("lateral_conv0", x0, c1, 1, 1, false, &fpn_out0_dst);
  b.upsample(fpn_out0, Builder::slice(cat_p4, 0, c1));
  View f_out0 = b.csp("C3_p4", cat_p4, c1, n, false);
  View fpn_out1_dst = Builder::slice(cat_n3, c0, c0);
  View fpn_out1 = b.base_conv("reduce_conv1", f_out0, c0, 1, 1, false, &fpn_out1_dst);
  b.upsample(fpn_out1, Builder::slice(cat_p3, 0, c0));
  View pan_out2 = b.csp("C3_p3", cat_p3, c0, n, false);
  View bu2_dst = Builder::slice(cat_n3, 0, c0);
  b.conv("bu_conv2", pan_out2, c0, 3, 2, &bu2_dst);
  View pan_out1 = b.csp("C3_n3", cat_n3, c1, n, false);
  View bu1_dst = Builder::slice(cat_n4, 0, c1);
  b.conv("bu_conv1", pan_out1, c1, 3, 2, &bu1_dst);
  View pan_out0 = b.csp("C3_n4", cat_n4, c2, n, false);
  net.fpn[0] = pan_out2; net.fpn[1] = pan_out1; net.fpn[2] = pan_out0;

  net.buf_off.resize(net.bufs.size());
  size_t off = 0;
  for (size_t i = 0; i < net.bufs.size(); ++i) {
    net.buf_off[i] = off;
    off += (net.bufs[i].per_image() + 63) / 64 * 64;   // keep every buffer 256-B aligned
  }
  net.per_image_floats = off;
  net.tab_off.resize(net.bufs.size());
  int toff = 0;
  for (size_t i = 0; i < net.bufs.size(); ++i) { net.tab_off[i] = toff; toff += net.bufs[i].C; }
  net.tab_channels = (toff + 3) / 4 * 4;
  int soff = 0;
  for (auto& cw : net.convs) { cw.stat_off = soff; soff += cw.cout; }
  net.stat_channels = soff;

  // Backward bookkeeping: walking the ops in reverse, the first contributor to a gradient view
  // writes it, later ones accumulate.  The three FPN outputs are seeded from outside first.
  std::vector<std::vector<char>> written(net.bufs.size());
  for (size_t i = 0; i < net.bufs.size(); ++i) written[i].assign(net.bufs[i].C, 0);
  auto mark = [&](const View& v, bool& acc, const char* what, const std::string& name) -> int {
    int n_w = 0;
    for (int c = 0; c < v.C; ++c) n_w += written[v.buf][v.coff + c];
    JN_CHECK(n_w == 0 || n_w == v.C, JN_EINVAL, "backward plan: partially written gradient view (%s of %s)", what,
             name.c_str());
    acc = n_w == v.C;
    for (int c = 0; c < v.C; ++c) written[v.buf][v.coff + c] = 1;
    return JN_OK;
  };
  for (int i = 0; i < 3; ++i) { bool dummy; int rc = mark(net.fpn[i], dummy, "fpn", "seed"); if (rc) return rc; }
  for (auto it = net.ops.rbegin(); it != net.ops.rend(); ++it) {
    Op& op = *it;
    int rc = JN_OK;
    switch (op.kind) {
      case OP_STEM: case OP_PRED: break;
      case OP_SPP: op.acc_in = true; break;                       // adds into slice 0, written by the cat consumer
      case OP_ADDACT:
        if ((rc = mark(op.in, op.acc_in, "in", op.name))) return rc;
        if ((rc = mark(op.res, op.acc_res, "res", op.name))) return rc;
        break;
      default:
        if ((rc = mark(op.in, op.acc_in, "in", op.name))) return rc;
    }
  }
  return JN_OK;
}

// YOLOXHead (inference branch) appended to the detector's op list: per level stem 1x1, two 3x3
// convs per branch, and one OP_PRED (the three predictor convs + decode).  State-dict names
// follow upstream ("yolox.head.stems.0.conv.weight", "yolox.head.cls_preds.0.bias", ...).
int build_head(Net& net, std::vector<ParamEntry>& params, const std::string& prefix, float width, bool depthwise,
               int num_classes) {
  JN_CHECK(num_classes == 1, JN_EINVAL, "the needle detector has one class");
  Builder b{net, params, prefix};
  net.n_backbone_ops = (int)net.ops.size();
  const int hid = (int)(256 * width);
  const int strides[3] = {8, 16, 32};
  int a0 = 0;
  for (int k = 0; k < 3; ++k) {
    const std::string s = std::to_string(k);
    View x = b.base_conv("stems." + s, net.fpn[k], hid, 1, 1, false);
    View c = b.conv("cls_convs." + s + ".0", x, hid, 3, 1);
    c = b.conv("cls_convs." + s + ".1", c, hid, 3, 1);
    View r = b.conv("reg_convs." + s + ".0", x, hid, 3, 1);
    r = b.conv("reg_convs." + s + ".1", r, hid, 3, 1);
    Op op; op.kind = OP_PRED; op.in = r; op.res = c; op.act = ACT_NONE; op.name = "preds." + s;
    op.stride = strides[k]; op.level = k; op.anchor0 = a0;
    b.add_param(prefix + "cls_preds." + s + ".weight", {num_classes, hid, 1, 1}, 0, false, true);
    b.add_param(prefix + "cls_preds." + s + ".bias", {num_classes}, 0, false, true);
    b.add_param(prefix + "reg_preds." + s + ".weight", {4, hid, 1, 1}, 0, false, true);
    b.add_param(prefix + "reg_preds." + s + ".bias", {4}, 0, false, true);
    b.add_param(prefix + "obj_preds." + s + ".weight", {1, hid, 1, 1}, 0, false, true);
    b.add_param(prefix + "obj_preds." + s + ".bias", {1}, 0, false, true);
    net.ops.push_back(op);
    a0 += r.H * r.W;
  }
  net.n_anchors = a0;
  net.head_hid = hid;
  // backward bookkeeping of the head ops (detector training): the head is differentiated before the PAFPN, its stems
  // are the first writers of the three FPN gradient views (which the PAFPN plan treats as seeded from outside)
  {
    std::vector<std::vector<char>> written(net.bufs.size());
    for (size_t i = 0; i < net.bufs.size(); ++i) written[i].assign(net.bufs[i].C, 0);
    for (int oi = (int)net.ops.size() - 1; oi >= net.n_backbone_ops; --oi) {
      Op& op = net.ops[oi];
      if (op.kind == OP_PRED) {       // the predictor backward writes g[reg_feat] and g[cls_feat] in full
        for (int c = 0; c < op.in.C; ++c) written[op.in.buf][op.in.coff + c] = 1;
        for (int c = 0; c < op.res.C; ++c) written[op.res.buf][op.res.coff + c] = 1;
        continue;
      }
      int n_w = 0;
      for (int c = 0; c < op.in.C; ++c) n_w += written[op.in.buf][op.in.coff + c];
      JN_CHECK(n_w == 0 || n_w == op.in.C, JN_EINVAL, "backward plan: partially written gradient view (head op %s)", op.name.c_str());
      op.acc_in = n_w == op.in.C;
      for (int c = 0; c < op.in.C; ++c) written[op.in.buf][op.in.coff + c] = 1;
    }
  }
  // buffer / table / stats offsets grew with the head
  net.buf_off.resize(net.bufs.size());
  size_t off = 0;
  for (size_t i = 0; i < net.bufs.size(); ++i) { net.buf_off[i] = off; off += (net.bufs[i].per_image() + 63) / 64 * 64; }
  net.per_image_floats = off;
  net.tab_off.resize(net.bufs.size());
  int toff = 0;
  for (size_t i = 0; i < net.bufs.size(); ++i) { net.tab_off[i] = toff; toff += net.bufs[i].C; }
  net.tab_channels = (toff + 3) / 4 * 4;
  int soff = 0;
  for (auto& cw : net.convs) { cw.stat_off = soff; soff += cw.cout; }
  net.stat_channels = soff;
  return JN_OK;
}

}  // namespace jnr
