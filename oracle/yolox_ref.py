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
    """Inference branch only: raw maps -> [B, A, 5+nc] decoded to pixels."""

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
    """Inference branch of NeedleYOLOX.forward (src/models/yolox.py:24-57, 74-113)."""

    def __init__(self, backbone, head, conf_threshold: float):
        super().__init__(backbone, head)
        self.conf_threshold = conf_threshold

    def forward(self, patches: torch.Tensor, targets=None):
        assert targets is None, "oracle restates the inference branch only"
        mode = self.training
        fpn_outs = self.backbone(patches)          # current mode (yolox.py:54-55)
        self.eval()                                # yolox.py:77
        outputs = self.head(fpn_outs)
        outputs = postprocess(outputs, num_classes=1, class_agnostic=True,
                              conf_thre=self.conf_threshold)
        for b in outputs:                          # clamp_outputs, yolox.py:93-113
            if b is not None:
                b[:, :4].clamp_(min=0, max=patches.shape[-1] - 1)
        self.train(mode)
        return outputs, fpn_outs, {}
