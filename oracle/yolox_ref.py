"""Oracle restatement of the YOLOX networks the reference imports.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference only *imports* this code (``from yolox.models import yolox_nano ...``,
src/models/gpt.py:24, 242-264; ``YOLOX, YOLOXHead, YOLOPAFPN, postprocess``,
src/models/yolox.py:7-10).  The package itself (pierrot-lc/YOLOX fork of
Megvii-BaseDetection/YOLOX, no commit pinned, README.md:24-30) is not under
/root/reference, so this file restates the published topology.  **Parity
unpinned**: validated only by parameter counts (nano 0.912 M / tiny 5.056 M /
s 8.968 M at 80 classes), output shapes and state-dict key names.

Module/attribute names follow upstream so that ``state_dict()`` keys are the
ones reference checkpoints hold (``backbone.backbone.dark2.0.dconv.conv.weight``
...).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# (depth, width, depthwise) per model name, as selected at src/models/gpt.py:242-250
YOLOX_SIZES = {
    "yolox-nano": (0.33, 0.25, True),
    "yolox-tiny": (0.33, 0.375, False),
    "yolox-s": (0.33, 0.50, False),
    "yolox-m": (0.67, 0.75, False),
    "yolox-l": (1.0, 1.0, False),
    "yolox-x": (1.33, 1.25, False),
}
YOLOX_SIZES["yolox"] = YOLOX_SIZES["yolox-nano"]

BN_EPS = 1e-3
BN_MOMENTUM = 0.03


class BaseConv(nn.Module):
    """Conv2d(bias=False, pad=(k-1)//2) -> BatchNorm2d -> SiLU."""

    def __init__(self, cin, cout, ksize, stride, groups=1):
        super().__init__()
        pad = (ksize - 1) // 2
        self.conv = nn.Conv2d(cin, cout, ksize, stride, pad, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOMENTUM)
        self.act = nn.SiLU(inplace=False)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class DWConv(nn.Module):
    """Depthwise ksize conv followed by pointwise 1x1 (each with BN + SiLU)."""

    def __init__(self, cin, cout, ksize, stride=1):
        super().__init__()
        self.dconv = BaseConv(cin, cin, ksize, stride, groups=cin)
        self.pconv = BaseConv(cin, cout, 1, 1)

    def forward(self, x):
        return self.pconv(self.dconv(x))


def _conv(depthwise):
    return DWConv if depthwise else BaseConv


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, shortcut=True, expansion=0.5, depthwise=False):
        super().__init__()
        hidden = int(cout * expansion)
        self.conv1 = BaseConv(cin, hidden, 1, 1)
        self.conv2 = _conv(depthwise)(hidden, cout, 3, 1)
        self.use_add = shortcut and cin == cout

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.use_add else y


class SPPBottleneck(nn.Module):
    def __init__(self, cin, cout, kernel_sizes=(5, 9, 13)):
        super().__init__()
        hidden = cin // 2
        self.conv1 = BaseConv(cin, hidden, 1, 1)
        self.m = nn.ModuleList(
            [nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2) for k in kernel_sizes]
        )
        self.conv2 = BaseConv(hidden * (len(kernel_sizes) + 1), cout, 1, 1)

    def forward(self, x):
        x = self.conv1(x)
        x = torch.cat([x] + [m(x) for m in self.m], dim=1)
        return self.conv2(x)


class CSPLayer(nn.Module):
    def __init__(self, cin, cout, n=1, shortcut=True, expansion=0.5, depthwise=False):
        super().__init__()
        hidden = int(cout * expansion)
        self.conv1 = BaseConv(cin, hidden, 1, 1)
        self.conv2 = BaseConv(cin, hidden, 1, 1)
        self.conv3 = BaseConv(2 * hidden, cout, 1, 1)
        self.m = nn.Sequential(
            *[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise) for _ in range(n)]
        )

    def forward(self, x):
        x1 = self.m(self.conv1(x))
        x2 = self.conv2(x)
        return self.conv3(torch.cat((x1, x2), dim=1))


class Focus(nn.Module):
    """Space-to-depth (TL, BL, TR, BR) then a dense ksize conv."""

    def __init__(self, cin, cout, ksize=1, stride=1):
        super().__init__()
        self.conv = BaseConv(cin * 4, cout, ksize, stride)

    def forward(self, x):
        tl = x[..., ::2, ::2]
        tr = x[..., ::2, 1::2]
        bl = x[..., 1::2, ::2]
        br = x[..., 1::2, 1::2]
        return self.conv(torch.cat((tl, bl, tr, br), dim=1))


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, depthwise=False):
        super().__init__()
        Conv = _conv(depthwise)
        bc = int(wid_mul * 64)
        bd = max(round(dep_mul * 3), 1)
        self.stem = Focus(3, bc, ksize=3)
        self.dark2 = nn.Sequential(
            Conv(bc, bc * 2, 3, 2),
            CSPLayer(bc * 2, bc * 2, n=bd, depthwise=depthwise),
        )
        self.dark3 = nn.Sequential(
            Conv(bc * 2, bc * 4, 3, 2),
            CSPLayer(bc * 4, bc * 4, n=bd * 3, depthwise=depthwise),
        )
        self.dark4 = nn.Sequential(
            Conv(bc * 4, bc * 8, 3, 2),
            CSPLayer(bc * 8, bc * 8, n=bd * 3, depthwise=depthwise),
        )
        self.dark5 = nn.Sequential(
            Conv(bc * 8, bc * 16, 3, 2),
            SPPBottleneck(bc * 16, bc * 16),
            CSPLayer(bc * 16, bc * 16, n=bd, shortcut=False, depthwise=depthwise),
        )

    def forward(self, x):
        x = self.stem(x)
        x = self.dark2(x)
        d3 = self.dark3(x)
        d4 = self.dark4(d3)
        d5 = self.dark5(d4)
        return d3, d4, d5


class YOLOPAFPN(nn.Module):
    def __init__(self, depth=1.0, width=1.0, in_channels=(256, 512, 1024), depthwise=False):
        super().__init__()
        self.backbone = CSPDarknet(depth, width, depthwise=depthwise)
        Conv = _conv(depthwise)
        c0, c1, c2 = (int(c * width) for c in in_channels)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise)
        self.bu_conv2 = Conv(c0, c0, 3, 2)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise)
        self.bu_conv1 = Conv(c1, c1, 3, 2)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise)

    def forward(self, x):
        x2, x1, x0 = self.backbone(x)
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(torch.cat([self.upsample(fpn_out0), x1], 1))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(torch.cat([self.upsample(fpn_out1), x2], 1))
        pan_out1 = self.C3_n3(torch.cat([self.bu_conv2(pan_out2), fpn_out1], 1))
        pan_out0 = self.C3_n4(torch.cat([self.bu_conv1(pan_out1), fpn_out0], 1))
        return (pan_out2, pan_out1, pan_out0)


class YOLOXHead(nn.Module):
    """Inference branch (raw maps -> [B, A, 5+nc] decoded to pixels) and the training branch
    (`losses`: SimOTA assignment + IoU / objectness / class / L1 losses, restated from the published YOLOX v0.3.0
    yolo_head.py — the fork the reference installs is not vendored and pins no commit: PARITY UNPINNED)."""

    def __init__(self, num_classes, width=1.0, strides=(8, 16, 32),
                 in_channels=(256, 512, 1024), depthwise=False):
        super().__init__()
        self.num_classes = num_classes
        self.strides = list(strides)
        Conv = _conv(depthwise)
        hid = int(256 * width)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.cls_preds = nn.ModuleList()
        self.reg_preds = nn.ModuleList()
        self.obj_preds = nn.ModuleList()
        self.stems = nn.ModuleList()
        for c in in_channels:
            self.stems.append(BaseConv(int(c * width), hid, 1, 1))
            self.cls_convs.append(nn.Sequential(Conv(hid, hid, 3, 1), Conv(hid, hid, 3, 1)))
            self.reg_convs.append(nn.Sequential(Conv(hid, hid, 3, 1), Conv(hid, hid, 3, 1)))
            self.cls_preds.append(nn.Conv2d(hid, num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hid, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hid, 1, 1, 1, 0))

    def raw_maps(self, feats):
        outs = []
        for k, x in enumerate(feats):
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            reg_feat = self.reg_convs[k](x)
            cls_out = self.cls_preds[k](cls_feat)
            reg_out = self.reg_preds[k](reg_feat)
            obj_out = self.obj_preds[k](reg_feat)
            outs.append(torch.cat([reg_out, obj_out.sigmoid(), cls_out.sigmoid()], 1))
        return outs

    def forward(self, feats):
        outs = self.raw_maps(feats)
        hw = [o.shape[-2:] for o in outs]
        out = torch.cat([o.flatten(start_dim=2) for o in outs], dim=2).permute(0, 2, 1)
        return decode_outputs(out, hw, self.strides)

    # ---- training branch (YOLOXHead.forward(xin, labels, imgs) in train mode -> get_losses) -----------------
    def raw_logits(self, feats):
        """[B, A, 6] raw predictor outputs (reg 4, obj logit, cls logit), anchors ordered level by level, row-major."""
        outs = []
        for k, x in enumerate(feats):
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            reg_feat = self.reg_convs[k](x)
            o = torch.cat([self.reg_preds[k](reg_feat), self.obj_preds[k](reg_feat), self.cls_preds[k](cls_feat)], 1)
            outs.append(o.flatten(start_dim=2).permute(0, 2, 1))
        hw = [(f.shape[-2], f.shape[-1]) for f in feats]
        return torch.cat(outs, 1), hw

    def losses(self, feats, labels: torch.Tensor, use_l1: bool = True):
        """labels [B, nb, 5] = (class, cx, cy, w, h), zero rows = padding.  Returns the reference's tuple
        (loss, 5 * iou_loss, obj_loss, cls_loss, l1_loss, num_fg / max(num_gts, 1))."""
        assert self.num_classes == 1
        raw, hw = self.raw_logits(feats)
        B, A, _ = raw.shape
        grids, svec = [], []
        for (h, w), s in zip(hw, self.strides):
            yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            grids.append(torch.stack((xv, yv), 2).view(-1, 2))
            svec.append(torch.full((h * w,), float(s)))
        grid = torch.cat(grids, 0).to(raw.dtype)
        stride = torch.cat(svec, 0).to(raw.dtype)
        xy = (raw[..., 0:2] + grid) * stride[:, None]
        wh = torch.exp(raw[..., 2:4]) * stride[:, None]
        boxes = torch.cat([xy, wh], -1)                       # decoded cxcywh predictions (with grad)
        obj, cls = raw[..., 4], raw[..., 5]
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)            # rows with a positive sum count as objects
        loss_iou = raw.new_zeros(())
        loss_cls = raw.new_zeros(())
        loss_l1 = raw.new_zeros(())
        obj_target = torch.zeros((B, A), dtype=raw.dtype)
        num_fg, num_gts = 0.0, 0.0
        assign = []
        for b in range(B):
            ng = int(nlabel[b])
            num_gts += ng
            if ng == 0:
                assign.append(None)
                continue
            gt = labels[b, :ng, 1:5].to(raw.dtype)            # the FIRST ng rows (as published, even if a zero row precedes)
            with torch.no_grad():
                fg, matched, ious = simota_assign(gt, boxes[b].detach(), obj[b].detach(), cls[b].detach(), grid, stride)
            assign.append((fg, matched, ious))
            nf = int(fg.sum())
            num_fg += nf
            if nf == 0:
                continue
            obj_target[b, fg] = 1.0
            tgt = gt[matched]
            pb = boxes[b][fg]
            loss_iou = loss_iou + iou_loss(pb, tgt).sum()
            loss_cls = loss_cls + F.binary_cross_entropy_with_logits(cls[b][fg], ious, reduction="sum")
            if use_l1:
                l1t = torch.stack([tgt[:, 0] / stride[fg] - grid[fg, 0], tgt[:, 1] / stride[fg] - grid[fg, 1],
                                   torch.log(tgt[:, 2] / stride[fg] + 1e-8), torch.log(tgt[:, 3] / stride[fg] + 1e-8)], 1)
                loss_l1 = loss_l1 + (raw[b][fg][:, :4] - l1t).abs().sum()
        den = max(num_fg, 1.0)
        loss_obj = F.binary_cross_entropy_with_logits(obj, obj_target, reduction="sum") / den
        loss_iou, loss_cls, loss_l1 = loss_iou / den, loss_cls / den, loss_l1 / den
        total = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
        self.last_assignment = assign
        return total, 5.0 * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / max(num_gts, 1.0)


def pairwise_iou_cxcywh(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bboxes_iou(a, b, xyxy=False) of YOLOX: [len(a), len(b)]."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[None, :, :2] - b[None, :, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[None, :, :2] + b[None, :, 2:] / 2)
    area_a = a[:, 2] * a[:, 3]
    area_b = b[:, 2] * b[:, 3]
    en = (tl < br).to(a.dtype).prod(dim=2)
    inter = (br - tl).prod(dim=2) * en
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def iou_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """IOUloss(reduction="none", loss_type="iou"): 1 - iou^2 on cxcywh rows."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = pred[:, 2] * pred[:, 3]
    area_g = target[:, 2] * target[:, 3]
    en = (tl < br).to(pred.dtype).prod(dim=1)
    inter = (br - tl).prod(dim=1) * en
    iou = inter / (area_p + area_g - inter + 1e-16)
    return 1.0 - iou ** 2


def simota_assign(gt: torch.Tensor, boxes: torch.Tensor, obj: torch.Tensor, cls: torch.Tensor,
                  grid: torch.Tensor, stride: torch.Tensor):
    """get_assignments + get_geometry_constraint + simota_matching for one image (num_classes = 1).
    Returns (fg mask [A], matched gt index per fg anchor, IoU of each fg anchor with its gt)."""
    A, ng = boxes.shape[0], gt.shape[0]
    xc = (grid[:, 0] + 0.5) * stride
    yc = (grid[:, 1] + 0.5) * stride
    r = 1.5 * stride
    deltas = torch.stack([xc[None] - (gt[:, 0:1] - r[None]), yc[None] - (gt[:, 1:2] - r[None]),
                          (gt[:, 0:1] + r[None]) - xc[None], (gt[:, 1:2] + r[None]) - yc[None]], 2)
    in_centers = deltas.min(dim=-1).values > 0.0                # [ng, A]
    cand = in_centers.sum(0) > 0                                # anchor_filter
    fg = torch.zeros(A, dtype=torch.bool)
    if int(cand.sum()) == 0:
        return fg, torch.zeros((0,), dtype=torch.long), boxes.new_zeros((0,))
    geo = in_centers[:, cand]
    ious = pairwise_iou_cxcywh(gt, boxes[cand])                 # [ng, nc]
    iou_cost = -torch.log(ious + 1e-8)
    p = (cls[cand].sigmoid() * obj[cand].sigmoid()).sqrt()
    cls_cost = F.binary_cross_entropy(p[None, :].expand(ng, -1), torch.ones((ng, p.shape[0]), dtype=p.dtype),
                                      reduction="none")        # one class: the one-hot target is 1
    cost = cls_cost + 3.0 * iou_cost + 1e6 * (~geo).to(ious.dtype)
    match = torch.zeros_like(cost, dtype=torch.uint8)
    k10 = min(10, ious.shape[1])
    dyn_k = torch.clamp(torch.topk(ious, k10, dim=1).values.sum(1).int(), min=1)
    for g in range(ng):
        idx = torch.topk(cost[g], k=int(dyn_k[g]), largest=False).indices
        match[g, idx] = 1
    per_anchor = match.sum(0)
    if int(per_anchor.max()) > 1:
        multi = per_anchor > 1
        amin = torch.min(cost[:, multi], dim=0).indices
        match[:, multi] = 0
        match[amin, multi] = 1
    sel = per_anchor > 0
    fg[cand.nonzero().squeeze(1)[sel]] = True
    matched = match[:, sel].argmax(0)
    pred_ious = (match.to(ious.dtype) * ious).sum(0)[sel]
    return fg, matched, pred_ious


def decode_outputs(out: torch.Tensor, hw, strides) -> torch.Tensor:
    grids, svec = [], []
    for (h, w), s in zip(hw, strides):
        yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
        svec.append(torch.full((1, h * w, 1), float(s)))
    grids = torch.cat(grids, 1).to(out.dtype)
    svec = torch.cat(svec, 1).to(out.dtype)
    return torch.cat(
        [(out[..., 0:2] + grids) * svec, torch.exp(out[..., 2:4]) * svec, out[..., 4:]],
        dim=-1,
    )


def box_iou_xyxy(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_thr: float) -> torch.Tensor:
    """Greedy NMS as torchvision.ops.nms: keep order = descending score,
    suppress when IoU > thr.  Ties keep the lower original index first."""
    if boxes.numel() == 0:
        return torch.zeros((0,), dtype=torch.long)
    order = torch.sort(scores, descending=True, stable=True).indices
    iou = box_iou_xyxy(boxes[order], boxes[order])
    n = len(order)
    removed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(int(order[i]))
        removed |= iou[i] > iou_thr
    return torch.tensor(keep, dtype=torch.long)


def postprocess(prediction: torch.Tensor, num_classes: int, conf_thre: float = 0.7,
                nms_thre: float = 0.45, class_agnostic: bool = False
                ) -> List[Optional[torch.Tensor]]:
    """cxcywh->xyxy, obj*cls >= conf filter, NMS; rows are
    [x1, y1, x2, y2, obj_conf, class_conf, class_pred]; None when nothing survives."""
    assert class_agnostic, "reference only uses class_agnostic=True (src/models/yolox.py:80-85)"
    box = prediction.new_empty(prediction.shape)
    box[..., 0] = prediction[..., 0] - prediction[..., 2] / 2
    box[..., 1] = prediction[..., 1] - prediction[..., 3] / 2
    box[..., 2] = prediction[..., 0] + prediction[..., 2] / 2
    box[..., 3] = prediction[..., 1] + prediction[..., 3] / 2
    prediction = prediction.clone()
    prediction[..., :4] = box[..., :4]
    output: List[Optional[torch.Tensor]] = [None] * len(prediction)
    for i, pred in enumerate(prediction):
        if not pred.size(0):
            continue
        class_conf, class_pred = torch.max(pred[:, 5:5 + num_classes], 1, keepdim=True)
        conf_mask = (pred[:, 4] * class_conf.squeeze() >= conf_thre).squeeze()
        det = torch.cat((pred[:, :5], class_conf, class_pred.float()), 1)[conf_mask]
        if not det.size(0):
            continue
        keep = nms(det[:, :4], det[:, 4] * det[:, 5], nms_thre)
        output[i] = det[keep]
    return output


class YOLOX(nn.Module):
    def __init__(self, backbone: YOLOPAFPN, head: YOLOXHead):
        super().__init__()
        self.backbone = backbone
        self.head = head


def build_pafpn(name: str) -> YOLOPAFPN:
    depth, width, dw = YOLOX_SIZES[name]
    return YOLOPAFPN(depth, width, depthwise=dw)


def build_head(name: str, num_classes: int) -> YOLOXHead:
    _, width, dw = YOLOX_SIZES[name]
    return YOLOXHead(num_classes, width, depthwise=dw)


def build_yolox(name: str, num_classes: int) -> YOLOX:
    """The factories' topology (yolox_nano ... yolox_x); weights are torch default
    init (pretrained weights need the network, src/models/gpt.py:251-264)."""
    return YOLOX(build_pafpn(name), build_head(name, num_classes))


class NeedleYOLOXRef(YOLOX):
    """NeedleYOLOX.forward (src/models/yolox.py:24-113): inference branch, and with `targets` the loss branch
    (yolox.py:58-73: xyxy -> cxcywh, head in train mode, use_l1 = True)."""

    def __init__(self, backbone, head, conf_threshold: float):
        super().__init__(backbone, head)
        self.conf_threshold = conf_threshold

    def forward(self, patches: torch.Tensor, targets=None):
        mode = self.training
        fpn_outs = self.backbone(patches)          # current mode (yolox.py:54-55)
        losses = {}
        if targets is not None:
            t = targets.clone().to(patches.dtype)
            x1, y1, x2, y2 = t[..., 1].clone(), t[..., 2].clone(), t[..., 3].clone(), t[..., 4].clone()
            t[..., 1], t[..., 2], t[..., 3], t[..., 4] = (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1
            self.train()                           # yolox.py:62
            loss, iou_l, conf_l, cls_l, l1_l, num_fg = self.head.losses(fpn_outs, t, use_l1=True)
            losses = {"total_loss": loss, "iou_loss": iou_l, "l1_loss": l1_l, "conf_loss": conf_l, "cls_loss": cls_l,
                      "num_fg": num_fg}
        self.eval()                                # yolox.py:77
        outputs = self.head(fpn_outs)
        outputs = postprocess(outputs, num_classes=1, class_agnostic=True,
                              conf_thre=self.conf_threshold)
        for b in outputs:                          # clamp_outputs, yolox.py:93-113
            if b is not None:
                b[:, :4].clamp_(min=0, max=patches.shape[-1] - 1)
        self.train(mode)
        return outputs, fpn_outs, losses
