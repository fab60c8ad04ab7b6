"""Pins the CPU oracle against the golden vectors produced by the reference's own
code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import env_ref, rollout_ref, yolox_ref
from oracle.gpt_ref import _Block, build_gpt_ref, gelu_tanh
from oracle.posenc_ref import posenc2d_row_col, PositionalEncoding2D
from tests.helpers import randomize_bn, synth_batch, synth_tokens

T = torch.from_numpy


def test_g1_reference_test_env(golden):
    """tests/test_env.py:10-31 replayed: (1,0) RIGHT DOWN DOWN -> (3,1)."""
    g = golden("g1_env.npz")
    env = env_ref.EnvRef(torch.zeros(1, 3, 1792, 2240),
                         torch.tensor([[[310, 810, 400, 850], [700, 1500, 800, 1600]]]), 448, 8, 1)
    env.reset(torch.tensor([[1, 0]]))
    assert np.array_equal(env.bbox_masks.numpy(), g["t_env_bbox_masks"])
    for t, a in enumerate(g["t_env_actions"]):
        _, r, te, tr, info = env.step(torch.tensor([int(a)]))
        assert np.array_equal(r.numpy(), g["t_env_rewards"][:, t])
        assert np.array_equal(te.numpy(), g["t_env_terminated"][:, t])
        assert np.array_equal(info["positions"].numpy(), g["t_env_positions"][:, t])
    assert info["positions"].tolist() == [[3, 1]]
    assert np.array_equal(np.array(env_ref.ACTION_DELTAS_YX), g["action_deltas"])
    assert env_ref.n_action_classes(True) == int(g["nclasses_stop"])
    assert env_ref.n_action_classes(False) == int(g["nclasses_nostop"])


def test_g1_random_walks(golden):
    g = golden("g1_env.npz")
    for tag, stop in (("nostop", False), ("stop", True)):
        images, bboxes, start = T(g[f"{tag}_images"]), T(g[f"{tag}_bboxes"]), T(g[f"{tag}_start"])
        acts = T(g[f"{tag}_actions"])
        P = images.shape[-1] // 5
        env = env_ref.EnvRef(images, bboxes, P, acts.shape[1], 1, stop)
        assert np.array_equal(env.bbox_masks.numpy(), g[f"{tag}_bbox_masks"])
        p, info = env.reset(start)
        pats = [p]
        for t in range(acts.shape[1]):
            p, r, te, tr, info = env.step(acts[:, t])
            pats.append(p)
            assert np.array_equal(r.numpy(), g[f"{tag}_rewards"][:, t]), (tag, t)
            assert r.dtype == torch.float32
            assert np.array_equal(te.numpy(), g[f"{tag}_terminated"][:, t])
            assert np.array_equal(tr.numpy(), g[f"{tag}_truncated"][:, t])
            assert np.array_equal(info["positions"].numpy(), g[f"{tag}_positions"][:, t + 1])
            assert np.array_equal(env.visited_patches.numpy(), g[f"{tag}_visited"][:, t])
        assert np.array_equal(torch.cat(pats, 1).numpy(), g[f"{tag}_patches"])   # bit exact gather
        assert np.array_equal(env.prop_patches_found.numpy(), g[f"{tag}_prop_patches_found"])


def test_g7_known_answers(golden):
    g = golden("g7_known_answers.npz")
    env = env_ref.EnvRef(torch.zeros((1, 3, 1792, 2240)), T(g["targets_bboxes"]), 448, 20, 1)
    tg = env.get_detection_targets()
    assert len(tg) == 1 and np.array_equal(tg[0].numpy(), g["targets_expected"])


def test_g2_transformer(golden):
    g = golden("g2_transformer.npz")
    assert np.allclose(gelu_tanh(T(g["gelu_in"])).numpy(), g["gelu_out"], atol=1e-6)
    assert np.allclose(gelu_tanh(torch.tensor([-1.0, 0.0, 1.0])).numpy(), [-0.1588, 0.0, 0.8412], atol=1e-4)
    blk = _Block(48, 3, 9)
    blk.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")})
    with torch.no_grad():
        assert np.allclose(blk(T(g["block_in"])).numpy(), g["block_out"], atol=1e-5)
        assert np.allclose(blk.attn(T(g["block_in"])).numpy(), g["attn_out"], atol=1e-5)


def _g3_model(g):
    P, Tn = 64, 6
    m = build_gpt_ref(int(g["seed"]), patch_size=P, block_size=Tn, nclasses=9, with_detector=True,
                      image_processor="yolox-nano", gpt_backbone="yolox-nano")
    randomize_bn(m, int(g["bn_seed"]))
    return m.eval()


def test_g3_gpt_full_and_recurrent(golden):
    g = golden("g3_gpt_forward.npz")
    m = _g3_model(g)
    patches, actions, positions = synth_tokens(3, 6, 64, 9, 5, seed=int(g["tok_seed"]))
    assert np.array_equal(actions.numpy(), g["actions"])
    classes = torch.zeros(3, dtype=torch.long)
    with torch.no_grad():
        lg, emb = m(patches, actions, classes, positions)
        assert np.allclose(lg.numpy(), g["full_logits"], atol=1e-5)
        assert np.allclose(emb.numpy(), g["full_emb"], atol=1e-5)
        e, rec = None, []
        for t in range(6):
            l, e = m(patches[:, :t + 1], actions[:, :t + 1], classes, positions[:, :t + 1], e)
            rec.append(l[:, -1])
        assert np.allclose(torch.stack(rec, 1).numpy(), g["rec_logits"], atol=1e-5)
        assert np.allclose(e.numpy(), g["rec_emb"], atol=1e-5)
        # the recurrent path differs from the full one (1-D position 0 quirk, gpt.py:431-449)
        assert not np.allclose(g["rec_logits"], g["full_logits"], atol=1e-4)
        fpn = m.gpt_backbone(patches[:, 0])
        for i in range(3):
            assert np.allclose(fpn[i].numpy(), g[f"fpn{i}"], atol=1e-5)
        assert np.allclose(m.embed_patches(patches[:, :2]).numpy(), g["patch_emb"], atol=1e-5)


def test_g4_rollout_and_metrics(golden):
    g = golden("g4_rollout.npz")
    g3 = golden("g3_gpt_forward.npz")
    m = _g3_model(g3)
    P, Tn = int(g["P"]), int(g["T"])
    images, bboxes, _ = synth_batch(4, 4, 5, P, seed=int(g["batch_seed"]))
    assert np.array_equal(bboxes.numpy(), g["bboxes"])
    env = env_ref.EnvRef(images, bboxes, P, Tn, 1, True)
    with torch.no_grad():
        ro = rollout_ref.rollout(m, env, sample_actions=False, start_positions=T(g["start"]))
    for k in ("rewards", "returns", "logprobs", "entropies"):
        assert np.allclose(ro[k].numpy(), g[k], atol=1e-5), k
    for k in ("masks", "logit_masks", "positions"):
        assert np.array_equal(ro[k].numpy(), g[k]), k
    norm = rollout_ref.ReturnNormaliser()
    m1 = rollout_ref.reinforce_metrics(ro, 0.01, norm)
    norm.roll()
    m2 = rollout_ref.reinforce_metrics(ro, 0.01, norm)
    assert np.allclose(float(norm.mean), g["norm_mean"], atol=1e-6)
    assert np.allclose(float(norm.std), g["norm_std"], atol=1e-6)
    for tag, mm in (("m1", m1), ("m2", m2)):
        for k, v in mm.items():
            assert np.allclose(float(v), g[f"{tag}.{k}"], atol=1e-5), (tag, k)


def test_g6_yolox_self_checks():
    """Unpinned third-party topology: published parameter counts, shapes, key names."""
    def count(m):
        return sum(p.numel() for p in m.parameters())
    n = count(yolox_ref.build_yolox("yolox-nano", 80))
    t = count(yolox_ref.build_yolox("yolox-tiny", 80))
    s = count(yolox_ref.build_yolox("yolox-s", 80))
    assert round(n / 1e6, 3) == 0.912 and round(t / 1e6, 3) == 5.056 and round(s / 1e6, 3) == 8.968
    net = yolox_ref.build_pafpn("yolox-nano").eval()
    with torch.no_grad():
        f = net(torch.zeros(1, 3, 448, 448))
    assert [tuple(x.shape[1:]) for x in f] == [(64, 56, 56), (128, 28, 28), (256, 14, 14)]
    sd = net.state_dict()
    for k in ("backbone.stem.conv.conv.weight", "backbone.dark2.0.dconv.conv.weight",
              "backbone.dark2.1.m.0.conv2.pconv.bn.running_var", "backbone.dark5.1.conv2.conv.weight",
              "C3_n4.conv3.bn.num_batches_tracked", "lateral_conv0.conv.weight", "bu_conv1.dconv.conv.weight"):
        assert k in sd, k
    det = yolox_ref.NeedleYOLOXRef(yolox_ref.build_pafpn("yolox-nano"), yolox_ref.build_head("yolox-nano", 1), 0.0).eval()
    with torch.no_grad():
        raw = det.head(det.backbone(torch.zeros(1, 3, 448, 448)))
    assert tuple(raw.shape) == (1, 4116, 6)


def test_posenc_closed_form():
    pe = PositionalEncoding2D(48)
    table = pe(torch.zeros(1, 6, 4, 48))[0]                 # [x(col), y(row), C]
    rows = torch.tensor([0, 3, 1]); cols = torch.tensor([5, 0, 2])
    assert torch.allclose(table[cols, rows], posenc2d_row_col(rows, cols, 48))


def test_nms_and_postprocess_known_answers():
    boxes = torch.tensor([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert yolox_ref.nms(boxes, scores, 0.45).tolist() == [3, 2]
    pred = torch.zeros(1, 3, 6)
    pred[0, 0] = torch.tensor([5., 5, 10, 10, 0.9, 0.9])      # cx,cy,w,h,obj,cls
    pred[0, 1] = torch.tensor([5.5, 5.5, 10, 10, 0.8, 0.9])   # suppressed
    pred[0, 2] = torch.tensor([50., 50, 4, 4, 0.5, 0.5])      # below threshold 0.5 (0.25)
    out = yolox_ref.postprocess(pred, 1, conf_thre=0.5, class_agnostic=True)
    assert out[0].shape == (1, 7) and out[0][0, :4].tolist() == [0, 0, 10, 10]
    assert yolox_ref.postprocess(pred * 0, 1, conf_thre=0.5, class_agnostic=True) == [None]


def test_simota_known_answers():
    """Hand-computable cases of the restated SimOTA assignment (the YOLOX package is absent from the reference tree:
    parity unpinned, the oracle is checked against the published algorithm's invariants)."""
    from oracle.yolox_ref import simota_assign, iou_loss, pairwise_iou_cxcywh
    yv, xv = torch.meshgrid(torch.arange(4), torch.arange(4), indexing="ij")
    grid = torch.stack((xv, yv), 2).view(-1, 2).float()
    stride = torch.full((16,), 8.0)
    gt = torch.tensor([[12.0, 12.0, 8.0, 8.0]])
    boxes = gt.expand(16, 4).clone()                      # every anchor predicts the gt exactly: IoU 1
    fg, matched, ious = simota_assign(gt, boxes, torch.zeros(16), torch.zeros(16), grid, stride)
    # centre radius 1.5 * stride = 12 px around (12, 12): anchor centres 4, 12, 20 qualify -> 3 x 3 candidates;
    # dynamic k = int(sum of the top-10 IoUs) = 9 -> all of them are foreground
    assert int(fg.sum()) == 9 and fg.view(4, 4)[:3, :3].all() and matched.tolist() == [0] * 9
    assert torch.allclose(ious, torch.ones(9))
    boxes2 = boxes.clone()
    boxes2[:, 2:] = 4.0                                   # quarter-area predictions: IoU 0.25 -> k = int(2.25) = 2
    fg2, _, ious2 = simota_assign(gt, boxes2, torch.zeros(16), torch.zeros(16), grid, stride)
    assert int(fg2.sum()) == 2 and torch.allclose(ious2, torch.full((2,), 0.25))
    # two gts competing for the same anchors: every foreground anchor ends with exactly one gt (lowest cost)
    gt2 = torch.tensor([[12.0, 12.0, 8.0, 8.0], [14.0, 12.0, 8.0, 8.0]])
    fg3, matched3, _ = simota_assign(gt2, boxes, torch.zeros(16), torch.zeros(16), grid, stride)
    assert len(matched3) == int(fg3.sum()) and set(matched3.tolist()) <= {0, 1}
    assert torch.allclose(iou_loss(gt, gt), torch.zeros(1)) and float(pairwise_iou_cxcywh(gt, gt2)[0, 1]) == pytest.approx(0.6)


def test_philox_known_answers_and_dropout_masks():
    """Philox4x32-10 (the engine's counter-based generator: action sampling, env reset, dropout masks) against the
    known-answer vectors of Random123 (kat_vectors: philox4x32_10), and the keep-scale statistics of the dropout mask."""
    import numpy as np
    from oracle.dropout_ref import drop_scale, philox4x32

    def kat(ctr, key):
        r = philox4x32(key[0] | (key[1] << 32), *[np.array([c], dtype=np.uint32) for c in ctr])
        return [int(x[0]) for x in r]
    assert kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert kat((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    m = drop_scale(9, np.arange(8)[:, None, None], np.arange(21)[None, :, None], 2, 3, np.arange(192)[None, None, :], 0.1)
    assert set(np.unique(m).tolist()) == {0.0, np.float32(1.0 / 0.9).item()}
    assert abs((m == 0).mean() - 0.1) < 0.01
    assert not np.array_equal(m[0], m[1]) and not np.array_equal(m[:, 0], m[:, 1])
