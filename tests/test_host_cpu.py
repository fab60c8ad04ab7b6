"""CPU-only checks of the boundary: the C-ABI library loads and exports every declared
symbol, the state-dict table matches the oracle / reference names, host-side logic."""
import ctypes as C
import re
from pathlib import Path

import pytest
import numpy as np
import torch

import jolineedle_amd as ja
from jolineedle_amd import _lib
from jolineedle_amd.engine import Engine, make_jn_config
from tests.helpers import model_config

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    header = (ROOT / "include" / "jnroll.h").read_text()
    declared = set(re.findall(r"\b(jn_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.jn_abi_version() == _lib.ABI_VERSION == 2


def test_struct_layout_matches_header():
    header = (ROOT / "include" / "jnroll.h").read_text()
    body = header[header.index("typedef struct jn_config {"):header.index("} jn_config;")]
    fields = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip()
        m = re.match(r"(int32_t|float)\s+([^;]+);", line)
        if m:
            fields += [f.strip() for f in m.group(2).split(",")]
    assert fields == [f[0] for f in _lib.JnConfig._fields_]
    body = header[header.index("typedef struct jn_rollout_out {"):header.index("} jn_rollout_out;")]
    names = re.findall(r"\*\s*([a-z_]+_dev);", body)
    assert names == [f[0] for f in _lib.JnRolloutOut._fields_]


def test_create_rejects_bad_config_with_message():
    lib = _lib.load_library()
    cfg = make_jn_config(model_config(), 0, 4, 9)
    cfg.n_actions = 5
    h = C.c_void_p()
    assert lib.jn_create(C.byref(cfg), C.byref(h)) == -1
    assert b"n_actions" in lib.jn_last_error()
    cfg = make_jn_config(model_config(image_processor="yolox-tiny"), 0, 4, 9)
    assert lib.jn_create(C.byref(cfg), C.byref(h)) == -1      # 24-channel stem: not a multiple of 16
    cfg = make_jn_config(model_config(patch_size=100), 0, 4, 9)
    assert lib.jn_create(C.byref(cfg), C.byref(h)) == -1


@pytest.mark.parametrize("kw", [dict(), dict(gpt_backbone=None, image_processor="yolox-nano"),
                                dict(model_type="gpt-mini", patch_size=640, block_size=32, gpt_backbone="yolox-s"),
                                dict(concat_emb=False, decoder_pos_encoding=False, use_pos_emb=False, nclasses=8)])
def test_state_dict_table_matches_oracle(kw):
    """Key names + shapes are the reference's (load_state_dict(strict) both ways)."""
    from oracle.gpt_ref import build_gpt_ref
    okw = dict(kw)
    oracle = build_gpt_ref(1, **okw)
    m = ja.GPT(model_config(**kw), max_batch=2)
    sd, osd = m.state_dict(), oracle.state_dict()
    assert list(sorted(sd)) == list(sorted(osd))
    for k in sd:
        assert sd[k].shape == osd[k].shape and sd[k].dtype == osd[k].dtype, k
    m.load_state_dict(osd)
    oracle.load_state_dict(m.state_dict())
    names = {n for n, _ in m.named_parameters()}
    assert {n for n, _ in oracle.named_parameters()} == names
    # ... and in the reference's ORDER: torch.optim.AdamW keys its state by position in the parameter list, so the
    # "optimizer-gpt" / "optimizer-yolox" entries of a checkpoint name the same tensors on both sides only if the lists agree
    # (CSPLayer registers conv1, conv2, conv3, m; the engine's own table runs conv2|conv1 pairs and m before conv3)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in oracle.named_parameters()]
    assert list(sd) == list(osd)
    for prefix_is_yolox in (False, True):
        mine = [(n, tuple(p.shape)) for n, p in m.named_parameters() if n.startswith("yolox") == prefix_is_yolox]
        ref = [(n, tuple(p.shape)) for n, p in oracle.named_parameters() if n.startswith("yolox") == prefix_is_yolox]
        assert mine == ref               # the two optimiser groups of gpt.py:547-562, position by position
    g, y = m.configure_optimizers(ja.CfgNode(learning_rate=1e-4, yolo_lr=1e-4))
    n_gpt = sum(p.numel() for grp in g.param_groups for p in grp["params"])
    assert n_gpt == sum(p.numel() for n, p in oracle.named_parameters() if not n.startswith("yolox"))


def test_plan_has_the_surveyed_layer_counts():
    m = ja.GPT(model_config(), max_batch=1)
    tab = m.engine().param_table()
    nano_convs = [n for n, *_ in tab if n.startswith("gpt_backbone.") and n.endswith("conv.weight")]
    assert len(nano_convs) == 77
    s_convs = [n for n, *_ in tab if n.startswith("yolox.backbone.") and n.endswith("conv.weight")]
    assert len(s_convs) == 59


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = ja.GPT(model_config(), max_batch=1)
    with pytest.raises(_lib.JnError):
        m.sync_weights()
    with pytest.raises(RuntimeError):
        ja.NeedleGeneralEnv(torch.zeros(1, 3, 64, 64), torch.zeros(1, 1, 4, dtype=torch.long), 32, 4)
    with pytest.raises(_lib.LibraryNotBuilt):
        _lib.load_library("/nonexistent/libjnroll.so")


def test_action_vocabulary_and_args(golden):
    g = golden("g1_env.npz")
    deltas = [ja.ACTION_DELTAS[ja.Action(i)] for i in range(9)]
    assert deltas == [tuple(r) for r in g["action_deltas"].tolist()]
    assert ja.get_actions_info(ja.CfgNode(stop_enabled=True))[0].nclasses == int(g["nclasses_stop"])
    assert ja.get_actions_info(ja.CfgNode(stop_enabled=False))[0].nclasses == int(g["nclasses_nostop"])
    # the reference's own RL test command line (tests/test_rl.py:13-46) parses
    args = ja.get_args(["--seed", "12345", "--training-mode", "reinforce", "--model-type", "gpt-nano",
                        "--gpt-backbone", "yolox-nano", "--image-processor", "yolox-s", "--concat-embeddings",
                        "--decoder-pos-encoding", "--use-positional-embedding", "--max-seq-len", "8",
                        "--batch-size", "4", "--dropout", "0.0", "--patch-size", "448", "--devices", "0",
                        "--enable-stop", "--detector-conf-threshold", "0.50"])
    t, m = ja.args_to_config(args)
    assert (t.max_seq_len, t.stop_enabled, t.patch_size, m.pos_emb_size, m.block_size) == (8, True, 448, 25, 8)
    assert m.gpt_backbone == "yolox-nano" and m.concat_emb and m.decoder_pos_encoding


# --------------------------------------------------------------------------------------
# detection bookkeeping, pinned by the known answers of the reference's tests/test_map.py (G7)
# --------------------------------------------------------------------------------------
def test_detection_targets_and_patch_to_full_image_known_answers(golden):
    g = golden("g7_known_answers.npz")
    tg = ja.detection_targets(torch.from_numpy(g["targets_bboxes"]), 1792 // 448, 2240 // 448, 448)
    assert len(tg) == 1 and np.array_equal(tg[0].numpy(), g["targets_expected"])
    outs = [[torch.from_numpy(g["p2f_box0"]), torch.from_numpy(g["p2f_box1"]), None, torch.from_numpy(g["p2f_box3"])]]
    res = ja.patch_bboxes2full_image(outs, torch.from_numpy(g["p2f_offsets"]), torch.from_numpy(g["p2f_masks"]))
    assert len(res) == 1 and np.array_equal(res[0].numpy(), g["p2f_expected"])
    assert ja.patch_bboxes2full_image([[None, None]], torch.zeros((1, 2, 2), dtype=torch.long)) == [None]


def test_map50_known_answers(golden):
    """tests/test_map.py:36-66 of the reference: no prediction -> 0, perfect -> 1, one of five pieces missed -> 0.8."""
    g = golden("g7_known_answers.npz")
    targets = [torch.from_numpy(g["targets_expected"])]
    assert float(ja.compute_detection_metrics([None], targets)["map"]) == 0.0
    p2 = torch.tensor([[410, 410, 447, 446, 0.5, 1], [448, 410, 500, 447, 0.9, 1], [410, 448, 447, 500, 0.8, 1],
                       [448, 448, 500, 500, 0.7, 1], [1500, 1500, 1600, 1600, 0.6, 1]])
    assert float(ja.compute_detection_metrics([p2], targets)["map"]) == pytest.approx(1.0)
    p3 = p2[[0, 2, 3, 4]]
    assert float(ja.compute_detection_metrics([p3], targets)["map"]) == pytest.approx(0.8, 0.01)
    assert float(ja.compute_detection_metrics([p2], [torch.zeros((0, 5))])["map"]) == 0.0


def test_merge_boxes_groups_contiguous_pieces():
    pieces = torch.tensor([[410., 410, 447, 447, 0.9, 0.5], [448, 410, 500, 447, 0.8, 1.0], [410, 448, 447, 500, 0.7, 1.0],
                           [1500, 1500, 1600, 1600, 0.6, 1.0]])
    m = ja.merge_boxes(pieces, threshold=2)
    assert m.shape == (2, 6)
    assert m[0].tolist() == pytest.approx([410, 410, 500, 500, 0.8, 1.0])
    assert m[1].tolist() == pytest.approx([1500, 1500, 1600, 1600, 0.6, 1.0])
    tg = torch.tensor([[0, 410, 410, 447, 447], [0, 448, 410, 500, 447], [0, 900, 900, 950, 950]])
    assert ja.merge_boxes(tg, target=True).tolist() == [[0, 410, 410, 500, 447], [0, 900, 900, 950, 950]]
    assert ja.merge_boxes_batched([None, tg], target=True)[0] is None


def test_padded_collate_layout():
    """src/dataset.py:307-347: zero padding to the largest image rounded up to the patch size, zero-row padded boxes."""
    ims = [torch.ones((3, 100, 130)), 2 * torch.ones((3, 64, 200))]
    bbs = [torch.tensor([[1, 2, 30, 40]]), torch.tensor([[5, 5, 20, 20], [50, 10, 90, 60], [0, 0, 10, 10]])]
    b = ja.padded_collate(ims, bbs, 64)
    assert b["image"].shape == (2, 3, 128, 256) and b["bboxes"].shape == (2, 3, 4) and b["bboxes"].dtype == torch.long
    assert float(b["image"][0, :, :100, :130].min()) == 1.0 and float(b["image"][0, :, 100:].abs().sum()) == 0.0
    assert float(b["image"][1, :, :64, :200].min()) == 2.0 and float(b["image"][1, :, :, 200:].abs().sum()) == 0.0
    assert b["bboxes"][0].tolist() == [[1, 2, 30, 40], [0, 0, 0, 0], [0, 0, 0, 0]] and b["class_id"].tolist() == [0, 0]
    s1, s2 = ja.synthetic_batch(3, 4, 64, 7, device="cpu"), ja.synthetic_batch(3, 4, 64, 7, device="cpu")
    assert torch.equal(s1["image"], s2["image"]) and torch.equal(s1["bboxes"], s2["bboxes"])
    bb = s1["bboxes"]
    valid = bb.abs().sum(-1) > 0
    assert valid[:, 0].all() and (bb[valid][:, 2] <= 256).all() and (bb[valid][:, 2] > bb[valid][:, 0]).all()


# --------------------------------------------------------------------------------------
# detection augmentation: host-side parameter logic and the oracle's identities (no GPU)
# --------------------------------------------------------------------------------------
def test_augment_parameter_sampling_and_kernels():
    from jolineedle_amd.augment import DetectionAugment, NPARAM, gaussian_weights3, motion_kernel3
    w0, w1 = gaussian_weights3(torch.tensor([0.1, 0.7, 2.0]))
    assert torch.allclose(w0 + 2 * w1, torch.ones(3)) and float(w1[0]) < 1e-20 and (w1[1:] > 0).all() and (w0 > w1).all()
    for ang in (-180.0, -37.0, 0.0, 45.0, 90.0, 133.0):
        k = motion_kernel3(ang)
        assert abs(float(k.sum()) - 1.0) < 1e-6 and (k >= 0).all()
        assert torch.allclose(k, k.flip(0, 1), atol=1e-6)               # a line through the centre is point-symmetric
    assert torch.allclose(motion_kernel3(0.0)[1], torch.full((3,), 1 / 3))
    assert torch.allclose(motion_kernel3(90.0)[:, 1], torch.full((3,), 1 / 3), atol=1e-6)
    aug = DetectionAugment(planckian_coeffs=torch.tensor([[1.3, 0.7], [0.8, 1.2]]), seed=5)
    prm = aug.sample_params(4000)
    assert prm.shape == (4000, NPARAM)
    ident = torch.zeros(NPARAM); ident[0] = ident[1] = ident[3] = ident[10] = 1.0
    frac = lambda m: float(m.float().mean())
    assert abs(frac(prm[:, 0] != 1.0) - 0.5) < 0.04 and abs(frac(prm[:, 2] != 0) - 0.2) < 0.03
    assert abs(frac(prm[:, 4] != 0) - 0.5) < 0.04 and abs(frac(prm[:, 5] != 0) - 0.5) < 0.04
    assert abs(frac(prm[:, 10] != 1.0) - 0.3) < 0.04
    assert torch.allclose(prm[:, 3] + 2 * prm[:, 4], torch.ones(4000), atol=1e-6)
    assert torch.allclose(prm[:, 6:15].sum(1), torch.ones(4000), atol=1e-5)
    assert abs(frac(prm[:, 15] != 0) - 0.5) < 0.04                      # RandomPlasmaShadow(p=0.5)
    on = prm[:, 15] != 0
    assert (prm[on, 15] >= -0.2).all() and (prm[on, 15] < 0).all() and (prm[on, 16] >= 0).all() and (prm[on, 16] <= 0.4).all()
    assert (prm[on, 17] >= 0.1).all() and (prm[on, 17] <= 0.7).all() and (prm[on, 18] > 0.5).all()
    # the embedded CIE D-series table (kornia's "CIED" mode): warm illuminants boost red, cold ones blue, ~neutral at D65
    from jolineedle_amd.augment import planckian_cied_table
    tab = planckian_cied_table()
    assert tab.shape == (23, 2) and (tab[:, 0].diff() < 0).all() and (tab[:, 1].diff() > 0).all()
    assert abs(float(tab[5, 0]) - 1.0) < 0.03 and abs(float(tab[5, 1]) - 1.0) < 0.03       # 6500 K
    assert float(tab[0, 0]) > 1.2 and float(tab[0, 1]) < 0.7 and float(tab[-1, 1]) > 1.2
    dflt = DetectionAugment(seed=5).sample_params(400)
    assert abs(frac(dflt[:, 0] != 1.0) - 0.5) < 0.1                       # Planckian jitter works out of the box
    assert (DetectionAugment(p_planckian=0, p_gray=0, p_blur=0, p_noise=0, p_motion=0, p_shadow=0).sample_params(7) == ident).all()


def test_augment_oracle_identities():
    from jolineedle_amd.augment import NPARAM
    from oracle.augment_ref import augment_ref
    g = torch.Generator().manual_seed(2)
    x = torch.rand((2, 3, 12, 12), generator=g)
    nz = torch.randn((2, 3, 12, 12), generator=g)
    ident = torch.zeros((2, NPARAM)); ident[:, 0] = ident[:, 1] = ident[:, 3] = ident[:, 10] = 1.0
    assert torch.allclose(augment_ref(x, ident, nz), x, atol=1e-7)
    p = ident.clone(); p[:, 2] = 1.0
    y = augment_ref(x, p, nz)
    assert torch.allclose(y[:, 0], y[:, 1]) and torch.allclose(y[:, 0], 0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2], atol=1e-6)
    p = ident.clone(); p[:, 5] = 0.05
    assert torch.allclose(augment_ref(x, p, nz), x + 0.05 * nz, atol=1e-6)
    p = ident.clone(); p[:, 3], p[:, 4] = 0.5, 0.25                      # blur keeps a constant image constant (reflect border)
    assert torch.allclose(augment_ref(torch.full_like(x, 0.3), p, nz), torch.full_like(x, 0.3), atol=1e-6)
    # plasma shadow: a smooth fractal in [0, 1] around 0.5; the shaded fraction grows with the quantity; patches differ
    from oracle.augment_ref import plasma_ref
    from jolineedle_amd.augment import plasma_stretch
    st = float(plasma_stretch(torch.tensor([0.5]))[0])
    f0, f1 = plasma_ref(123, 0, 128, 0.5, st), plasma_ref(123, 1, 128, 0.5, st)
    assert f0.min() >= 0 and f0.max() <= 1 and not (f0 == f1).all()
    mean8 = np.mean([plasma_ref(7, k, 64, 0.5, st).mean() for k in range(8)])     # one patch is dominated by its 3 x 3 coarse lattice
    assert abs(float(mean8) - 0.5) < 0.12
    assert float(abs(f0[:, 1:] - f0[:, :-1]).mean()) < 0.05             # spatially smooth
    assert (f0 < 0.2).mean() < (f0 < 0.4).mean() < (f0 < 0.6).mean()
    xs = torch.full((1, 3, 128, 128), 0.5)
    p = torch.zeros((1, NPARAM)); p[:, 0] = p[:, 1] = p[:, 3] = p[:, 10] = 1.0
    p[:, 15], p[:, 16], p[:, 17], p[:, 18] = -0.2, 0.4, 0.5, st
    y = augment_ref(xs, p, torch.zeros_like(xs), seed=123)
    dark = (y - 0.4).abs() < 1e-6
    assert bool((dark | ((y - 0.5).abs() < 1e-6)).all()) and 0.02 < float(dark.float().mean()) < 0.9


# --------------------------------------------------------------------------------------
# config.json / infer helpers: host logic, no GPU
# --------------------------------------------------------------------------------------
def test_config_json_round_trip_and_infer_helpers(tmp_path):
    import json
    import jolineedle_amd as ja
    from tests.helpers import model_config
    mc = model_config(patch_size=64, block_size=4)
    tc = ja.CfgNode(patch_size=64, max_seq_len=4, stop_enabled=True, work_dir=str(tmp_path), env_name="run",
                    filter_classes={"plane", "runway"}, learning_rate=1e-4)
    path = ja.save_config(mc, tc)
    assert path == tmp_path / "run" / "config.json"                       # main.py:436-449 location
    cj = json.load(open(path))
    assert set(cj) == {"model", "train"} and cj["train"]["filter_classes"] in (["plane", "runway"], ["runway", "plane"])
    assert cj["model"]["actions_info"] == [{"action_type": "categorical", "nclasses": 9}]
    t2, m2 = ja.config_from_file(path)
    assert (t2.patch_size, t2.max_seq_len, t2.stop_enabled) == (64, 4, True)
    assert (m2.model_type, m2.gpt_backbone, m2.block_size) == ("gpt-nano", "yolox-nano", 4)
    assert not hasattr(m2, "actions_info")                                # rebuilt by the caller (infer.py:85-86)
    m2.actions_info = ja.get_actions_info(t2)
    assert m2.actions_info[0].nclasses == 9
    # image padding of infer.py:138-146 and the "cls x1 y1 x2 y2" target files
    x = torch.ones(1, 3, 100, 130)
    y = ja.pad_to_patch_multiple(x, 64)
    assert y.shape == (1, 3, 128, 192) and float(y[..., 100:, :].abs().sum()) == 0 and float(y[..., :, 130:].abs().sum()) == 0
    assert torch.equal(y[..., :100, :130], x) and ja.pad_to_patch_multiple(torch.ones(1, 3, 128, 64), 64).shape == (1, 3, 128, 64)
    f = tmp_path / "boxes.txt"
    f.write_text("0 10 20 30 40\n1 5 6 7 8 0.9\n\n")
    assert ja.load_bboxes(f) == [[10, 20, 30, 40], [5, 6, 7, 8]]


def test_supervised_compute_metrics_matches_the_reference_formula():
    """SupervisedTrainer.compute_metrics (src/supervised.py:138-198) is host-side torch arithmetic: weighted cross-entropy
    (weight[STOP] = stop_weight, reduction none) averaged over the non-padding tokens, accuracy over the same tokens, the
    detector's terms added under yolo_*, episode_length = mean number of real tokens."""
    from jolineedle_amd.supervised import SupervisedTrainer
    g = torch.Generator().manual_seed(3)
    B, T, nA = 3, 5, 9
    logits = torch.randn((B, T, nA), generator=g, requires_grad=True)
    actions = torch.randint(0, nA, (B, T), generator=g)
    actions[0, 2] = 8
    masks = torch.ones((B, T), dtype=torch.long); masks[1, 3:] = 0; masks[2, 1:] = 0
    tr = SupervisedTrainer.__new__(SupervisedTrainer)
    tr.config, tr.stop_weight = ja.CfgNode(stop_enabled=True), 0.25
    m = tr.compute_metrics(logits, actions, masks, {"total_loss": torch.tensor(0.5), "iou_loss": 0.125})
    w = torch.ones(nA); w[8] = 0.25
    keep = masks.flatten() == 1
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, nA), actions.flatten(), weight=w, reduction="none")[keep].mean()
    assert torch.allclose(m["action_loss"], ce) and torch.allclose(m["loss"], ce + 0.5)
    acc = (logits.reshape(-1, nA).argmax(1)[keep] == actions.flatten()[keep]).float().mean()
    assert torch.allclose(m["action_accuracy"], acc)
    assert float(m["episode_length"]) == float(masks.sum(1).float().mean())
    assert float(m["yolo_iou_loss"]) == 0.125 and float(m["yolo_loss"]) == 0.5
    m["loss"].backward()
    assert logits.grad is not None and float(logits.grad[2, 1:].abs().max()) == 0.0      # padding tokens carry no gradient


def test_every_library_switch_is_documented():
    """Every environment switch the library reads (``getenv("JN_…")`` in jolineedle_amd/csrc) is named in DESIGN.md — the
    switches are the A/B handles behind the measurements quoted there."""
    import glob
    import re
    root = Path(__file__).resolve().parents[1]
    names = set()
    for f in glob.glob(str(root / "jolineedle_amd" / "csrc" / "*")):
        if f.endswith((".hip", ".h", ".cpp")):
            names |= set(re.findall(r'getenv\("(JN_[A-Z0-9_]+)"\)', Path(f).read_text()))
    design = (root / "DESIGN.md").read_text()
    assert len(names) > 20
    assert not sorted(n for n in names if n not in design)
