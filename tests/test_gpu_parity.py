"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors.  Run on the MI355X box: python -m pytest tests -m gpu."""
import copy
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import jolineedle_amd as ja
from jolineedle_amd import _lib
from jolineedle_amd._lib import check, ptr
from tests.helpers import make_pair, model_config, randomize_bn, synth_batch, synth_tokens

pytestmark = pytest.mark.gpu
T_ = torch.from_numpy
DEV = "cuda:0"
# tolerance of BASELINE.json's north star: logits / boxes within 1e-3 of the CPU reference;
# fp32 activations + exact-fp32 MFMA keep the maps themselves to ~1e-4.
TOL_MAP = 2e-4
TOL_LOGIT = 1e-4


def _cfg(**kw):
    return ja.CfgNode(max_seq_len=kw.pop("T", 6), entropy_weight=0.01, stop_enabled=kw.pop("stop", True),
                      reward_norm=True, seed=1, **kw)


# --------------------------------------------------------------------------------------
# integer / byte work: bit exact
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,G,P", [(3, 3, (4, 5), 16), (2, 3, (2, 3), 448), (5, 3, (3, 3), 30), (1, 1, (1, 1), 7)])
def test_gather_bit_exact(B, C, G, P):
    g = torch.Generator().manual_seed(B * 100 + P)
    H, W = G[0] * P, G[1] * P
    images = torch.rand((B, C, H, W), generator=g).to(DEV)
    pos = torch.stack((torch.randint(0, G[0], (B,), generator=g), torch.randint(0, G[1], (B,), generator=g)), 1)
    out = torch.full((B, C, P, P), -1.0, device=DEV)
    lib = _lib.load_library()
    check(lib.jn_gather_patches(ptr(images), ptr(pos.to(DEV)), ptr(out), B, C, H, W, P, _lib.current_stream(torch.device(DEV))))
    torch.cuda.synchronize()
    ref = torch.stack([images[b, :, y * P:(y + 1) * P, x * P:(x + 1) * P] for b, (y, x) in enumerate(pos.tolist())])
    assert torch.equal(out, ref)


def test_gather_empty_batch():
    lib = _lib.load_library()
    x = torch.zeros(4, device=DEV)
    check(lib.jn_gather_patches(ptr(x), ptr(x), ptr(x), 0, 3, 16, 16, 16, None))


def test_env_reference_test_case(golden):
    """tests/test_env.py:10-31 of the reference through the device env."""
    g = golden("g1_env.npz")
    images = torch.zeros(1, 3, 1792, 2240, device=DEV)
    env = ja.NeedleGeneralEnv(images, torch.tensor([[[310, 810, 400, 850], [700, 1500, 800, 1600]]]), 448, 8, 1)
    patches, infos = env.reset(torch.tensor([[1, 0]]))
    assert torch.equal(infos["positions"].cpu(), torch.tensor([[1, 0]]))
    assert np.array_equal(env.bbox_masks.cpu().numpy(), g["t_env_bbox_masks"])
    for t, a in enumerate(g["t_env_actions"]):
        patches, r, te, tr, infos = env.step(torch.tensor([int(a)]))
        assert np.array_equal(r.cpu().numpy(), g["t_env_rewards"][:, t])
        assert np.array_equal(te.cpu().numpy(), g["t_env_terminated"][:, t])
    assert torch.equal(infos["positions"].cpu(), torch.tensor([[3, 1]]))
    assert patches.shape == (1, 1, 3, 448, 448)


@pytest.mark.parametrize("tag,stop", [("nostop", False), ("stop", True)])
def test_env_random_walks_golden(golden, tag, stop):
    """Reference env replays (sticky STOP, reward-before-visited, padding rows, clamping)."""
    g = golden("g1_env.npz")
    images, bboxes, start = T_(g[f"{tag}_images"]), T_(g[f"{tag}_bboxes"]), T_(g[f"{tag}_start"])
    acts = T_(g[f"{tag}_actions"])
    P = images.shape[-1] // 5
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, acts.shape[1], 1, stop)
    assert np.array_equal(env.bbox_masks.cpu().numpy(), g[f"{tag}_bbox_masks"])
    p, info = env.reset(start)
    pats = [p]
    for t in range(acts.shape[1]):
        p, r, te, tr, info = env.step(acts[:, t])
        pats.append(p)
        assert np.array_equal(r.cpu().numpy(), g[f"{tag}_rewards"][:, t]), (tag, t)
        assert np.array_equal(te.cpu().numpy(), g[f"{tag}_terminated"][:, t])
        assert np.array_equal(tr.cpu().numpy(), g[f"{tag}_truncated"][:, t])
        assert np.array_equal(info["positions"].cpu().numpy(), g[f"{tag}_positions"][:, t + 1])
        assert np.array_equal(env.visited_patches.cpu().numpy(), g[f"{tag}_visited"][:, t])
    assert np.array_equal(torch.cat(pats, 1).cpu().numpy(), g[f"{tag}_patches"])
    assert np.array_equal(env.prop_patches_found.cpu().numpy(), g[f"{tag}_prop_patches_found"])


def test_env_asserts_like_reference():
    with pytest.raises(AssertionError):
        ja.NeedleGeneralEnv(torch.zeros(1, 3, 100, 64, device=DEV), torch.zeros(1, 1, 4, dtype=torch.long), 32, 4)
    with pytest.raises(AssertionError):
        ja.NeedleGeneralEnv(torch.zeros(2, 3, 64, 64, device=DEV), torch.zeros(1, 1, 4, dtype=torch.long), 32, 4)


# --------------------------------------------------------------------------------------
# conv stack
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,N", [(64, 3), (96, 2), (448, 2)])
def test_nano_backbone_fpn_maps(P, N):
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None)
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(P))
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
    got = product.backbone_features(x)
    torch.cuda.synchronize()
    for i in range(3):
        assert got[i].shape == ref[i].shape
        err = (got[i].cpu() - ref[i]).abs().max().item()
        assert err < TOL_MAP, (i, err)


def test_dense_backbone_train_mode_batchnorm():
    """yolox-s encoder (dense 3x3 convs) in train mode: batch-statistics maps and running-stat updates."""
    product, oracle = make_pair(3, patch_size=96, block_size=6, with_detector=False, image_processor=None, gpt_backbone="yolox-s")
    x = torch.rand((3, 3, 96, 96), generator=torch.Generator().manual_seed(5))
    oracle.gpt_backbone.train()
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
    got = product.backbone_features(x, train=True)
    for i in range(3):
        assert (got[i].cpu() - ref[i]).abs().max().item() < 1e-3, i
    product.pull_bn_statistics()
    osd, psd = oracle.state_dict(), product.state_dict()
    for k in osd:
        if k.startswith("gpt_backbone") and ("running_mean" in k or "running_var" in k):
            assert torch.allclose(psd[k], osd[k], atol=1e-5, rtol=1e-4), k


@pytest.mark.parametrize("P,N", [(64, 5), (448, 3)])
def test_nano_backbone_train_mode_batchnorm(P, N):
    """Train-mode BN (batch statistics per pass, src/reinforce.py:304) + running-stat update."""
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None)
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(P + 1))
    oracle.gpt_backbone.train()
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
        ref = oracle.gpt_backbone(x)            # twice: running stats move twice
    product.backbone_features(x, train=True)
    got = product.backbone_features(x, train=True)
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        assert err < 1e-3, (i, err)          # batch statistics amplify fp32 rounding; north-star bound
    product.pull_bn_statistics()
    osd, psd = oracle.state_dict(), product.state_dict()
    for k in osd:
        if k.startswith("gpt_backbone") and ("running_mean" in k or "running_var" in k):
            assert torch.allclose(psd[k], osd[k], atol=1e-5, rtol=1e-4), k
    # back to eval: the refreshed running statistics are used
    oracle.gpt_backbone.eval()
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
    got = product.backbone_features(x)
    for i in range(3):
        assert (got[i].cpu() - ref[i]).abs().max().item() < 1e-3


@pytest.mark.parametrize("P,N,bb", [(64, 4, "yolox-nano"), (448, 2, "yolox-nano"), (64, 3, "yolox-s"), (160, 2, "yolox-s")])
def test_nano_backbone_backward_vs_autograd(P, N, bb):
    """Gradients of every conv weight and BN weight/bias of the patch encoder (train-mode BN)
    for loss = sum_i <fpn_i, R_i>, against torch autograd on the CPU oracle.  yolox-s = the dense 3x3
    (non-depthwise) encoder of BASELINE config 5: stride-1 / stride-2 data gradients and the 9-tap weight gradient."""
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None, gpt_backbone=bb)
    g = torch.Generator().manual_seed(17)
    x = torch.rand((N, 3, P, P), generator=g)
    net = oracle.gpt_backbone.train()
    outs = net(x)
    R = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * r).sum() for o, r in zip(outs, R))
    net.zero_grad()
    loss.backward()
    product.engine_zero_grad()
    product.backbone_features(x, train=True)
    product.backbone_backward(x, R)
    got = product.engine_grads("gpt_backbone.")
    torch.cuda.synchronize()
    worst = 0.0
    for name, p in net.named_parameters():
        gp = got["gpt_backbone." + name]
        ref = p.grad
        scale = ref.abs().max().item() + 1e-6
        err = (gp - ref).abs().max().item() / scale
        worst = max(worst, err)
        assert err < 2e-3, (name, err, scale)
    # accumulation: a second backward doubles the gradients
    product.backbone_features(x, train=True)
    product.backbone_backward(x, R)
    got2 = product.engine_grads("gpt_backbone.")
    for k in ("gpt_backbone.backbone.dark3.1.conv3.conv.weight", "gpt_backbone.backbone.dark3.0.conv.weight"):
        if k in got:
            assert torch.allclose(got2[k], 2 * got[k], rtol=1e-3, atol=1e-3 * got[k].abs().max().item())


def test_wide_data_gradient_on_the_bf16_pipe_matches_the_fp32_pipe(monkeypatch):
    """The data gradient of the wide 1x1 layers (256 / 512 channels: dark5, lateral_conv0, C3_p4 / C3_n4) runs on pw_x3_kernel
    (three bf16 planes per operand, transposed split weight) since round 4: every gradient of the encoder must stay where the
    fp32 matrix pipe (JN_NO_PW_X3_BWD=1, pw_dir_kernel<WT>) puts it — the layers below the wide ones see its output."""
    product, _ = make_pair(3, patch_size=64, block_size=6, with_detector=False, image_processor=None, gpt_backbone="yolox-nano")
    g = torch.Generator().manual_seed(23)
    x = torch.rand((6, 3, 64, 64), generator=g)
    R = [torch.randn(o.shape, generator=g) for o in product.backbone_features(x, train=True)]
    grads = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("JN_NO_PW_X3_BWD", "1")
        product.engine_zero_grad()
        product.backbone_features(x, train=True)
        product.backbone_backward(x, R)
        grads.append({k: v.clone() for k, v in product.engine_grads("gpt_backbone.").items()})
    torch.cuda.synchronize()
    worst = 0.0
    for k, a in grads[0].items():
        b = grads[1][k]
        rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
        worst = max(worst, rel)
        assert rel < 2e-5, (k, rel)
    assert worst > 0.0            # the two routes are different kernels: identical bits would mean the switch did nothing


def test_backbone_golden_and_patch_embedding(golden):
    g = golden("g3_gpt_forward.npz")
    product, oracle = make_pair(int(g["seed"]), int(g["bn_seed"]), patch_size=64, block_size=6,
                                image_processor="yolox-nano", gpt_backbone="yolox-nano")
    patches, actions, positions = synth_tokens(3, 6, 64, 9, 5, seed=int(g["tok_seed"]))
    fpn = product.backbone_features(patches[:, 0])
    for i in range(3):
        assert np.allclose(fpn[i].cpu().numpy(), g[f"fpn{i}"], atol=TOL_MAP), i
    emb = product.embed_patches(patches[:, :2])
    assert np.allclose(emb.cpu().numpy(), g["patch_emb"], atol=TOL_MAP)


# --------------------------------------------------------------------------------------
# decision model
# --------------------------------------------------------------------------------------
def test_gpt_forward_full_and_recurrent_golden(golden):
    """Reference GPT.forward outputs (full sequence and recurrent incl. the position-0 quirk)."""
    g = golden("g3_gpt_forward.npz")
    product, _ = make_pair(int(g["seed"]), int(g["bn_seed"]), patch_size=64, block_size=6,
                           image_processor="yolox-nano", gpt_backbone="yolox-nano")
    patches, actions, positions = synth_tokens(3, 6, 64, 9, 5, seed=int(g["tok_seed"]))
    classes = torch.zeros(3, dtype=torch.long)
    lg, emb = product(patches, actions, classes, positions)
    assert np.allclose(lg.cpu().numpy(), g["full_logits"], atol=TOL_LOGIT)
    assert np.allclose(emb.cpu().numpy(), g["full_emb"], atol=TOL_LOGIT)
    e, rec = None, []
    for t in range(6):
        l, e = product(patches[:, :t + 1], actions[:, :t + 1], classes, positions[:, :t + 1], e)
        assert l.shape == (3, t + 1, 9) and e.shape == (3, t + 2, 48)
        rec.append(l[:, -1])
    assert np.allclose(torch.stack(rec, 1).cpu().numpy(), g["rec_logits"], atol=TOL_LOGIT)
    assert np.allclose(e.cpu().numpy(), g["rec_emb"], atol=TOL_LOGIT)


@pytest.mark.parametrize("kw", [dict(concat_emb=False), dict(decoder_pos_encoding=False, nclasses=8),
                                dict(use_pos_emb=False), dict(model_type="gpt-mini"),
                                dict(model_type="gpt-mini", gpt_backbone="yolox-s"),     # BASELINE config 5 topology
                                dict(gpt_backbone=None, image_processor="yolox-nano"),
                                # the opt-in agent-batched MFMA step (kernels_gptmfma.hip), 4 and 16 agents per workgroup
                                dict(model_type="gpt-mini", _env=dict(JN_GPT_MFMA="1")),
                                dict(model_type="gpt-mini", concat_emb=False, _env=dict(JN_GPT_MFMA="1", JN_GPT_MFMA_AGENTS="16"))])
def test_gpt_forward_variants_vs_oracle(kw, monkeypatch):
    kw = dict(kw)
    for k, v in kw.pop("_env", {}).items():
        monkeypatch.setenv(k, v)
    base = dict(patch_size=64, block_size=5, image_processor="yolox-nano", gpt_backbone="yolox-nano")
    base.update(kw)
    product, oracle = make_pair(7, **base)
    nA = base.get("nclasses", 9)
    patches, actions, positions = synth_tokens(2, 5, 64, nA, 5, seed=11)
    classes = torch.tensor([37, 0])              # embed_class(classes), src/models/gpt.py:476-478 (class token of each sequence)
    with torch.no_grad():
        rl, re = oracle(patches, actions, classes, positions)
        assert (oracle(patches, actions, torch.zeros(2, dtype=torch.long), positions)[0] - rl).abs().max() > 10 * TOL_LOGIT
    lg, emb = product(patches, actions, classes, positions)
    assert (lg.cpu() - rl).abs().max() < TOL_LOGIT and (emb.cpu() - re).abs().max() < TOL_LOGIT
    with torch.no_grad():
        e_o, e_p = None, None
        for t in range(3):
            lo, e_o = oracle(patches[:, :t + 1], actions[:, :t + 1], classes, positions[:, :t + 1], e_o)
            lp, e_p = product(patches[:, :t + 1], actions[:, :t + 1], classes, positions[:, :t + 1], e_p)
            assert (lp.cpu() - lo).abs().max() < TOL_LOGIT


def test_gpt_forward_asserts_like_reference():
    product, _ = make_pair(7, patch_size=64, block_size=4, image_processor="yolox-nano")
    patches, actions, positions = synth_tokens(1, 5, 64, 9, 5, seed=1)
    with pytest.raises(AssertionError, match="Cannot forward sequence of length 5"):
        product(patches, actions, torch.zeros(1, dtype=torch.long), positions)
    with pytest.raises(AssertionError):
        product(patches[:, :2], actions[:, :2], torch.zeros(1, dtype=torch.long), None)
    with pytest.raises(AssertionError, match="class id outside embed_class"):      # nn.Embedding(100, C) raises on such an id
        product(patches[:, :2], actions[:, :2], torch.tensor([100]), positions[:, :2])


# --------------------------------------------------------------------------------------
# the rollout
# --------------------------------------------------------------------------------------
def test_rollout_greedy_golden(golden):
    """ReinforceTrainer.rollout of the reference (greedy) + compute_metrics, golden g4."""
    g = golden("g4_rollout.npz")
    g3 = golden("g3_gpt_forward.npz")
    P, Tn = int(g["P"]), int(g["T"])
    product, _ = make_pair(int(g3["seed"]), int(g3["bn_seed"]), patch_size=P, block_size=Tn,
                           image_processor="yolox-nano", gpt_backbone="yolox-nano")
    images, bboxes, _ = synth_batch(4, 4, 5, P, seed=int(g["batch_seed"]))
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    ro = tr.rollout(env, sample_actions=False, start_positions=T_(g["start"]))
    for k in ("masks", "logit_masks", "positions"):
        assert np.array_equal(ro[k].cpu().numpy(), g[k]), k
    assert np.array_equal(ro["rewards"].cpu().numpy(), g["rewards"])
    for k in ("returns", "logprobs", "entropies"):
        assert np.allclose(ro[k].cpu().numpy(), g[k], atol=TOL_LOGIT), k
    assert ro["patches"].shape == (4, g["masks"].shape[1], 3, P, P)
    pos = ro["positions"].cpu()
    for b in range(4):
        for t in range(pos.shape[1]):
            y, x = pos[b, t].tolist()
            assert torch.equal(ro["patches"][b, t].cpu(), images[b, :, y * P:(y + 1) * P, x * P:(x + 1) * P])
    m1 = tr.compute_metrics(ro)
    tr._compute_last_returns_mean_std()
    m2 = tr.compute_metrics(ro)
    assert np.allclose(float(tr.last_return_mean), g["norm_mean"], atol=1e-5)
    assert np.allclose(float(tr.last_return_std), g["norm_std"], atol=1e-5)
    for tag, mm in (("m1", m1), ("m2", m2)):
        for k, v in mm.items():
            assert np.allclose(float(v), g[f"{tag}.{k}"], atol=2e-4), (tag, k)


@pytest.mark.parametrize("stop,B,P,Tn", [(False, 5, 64, 7), (True, 3, 96, 5)])
def test_rollout_forced_vs_oracle(stop, B, P, Tn):
    from oracle import env_ref, rollout_ref
    nA = 9 if stop else 8
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, nclasses=nA, image_processor="yolox-nano")
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=31)
    forced = torch.randint(0, nA, (B, Tn), generator=torch.Generator().manual_seed(9))
    envo = env_ref.EnvRef(images, bboxes, P, Tn, 1, stop)
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, envo, forced_actions=forced, start_positions=start, stop_early=True)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, stop)
    ro = ja.ReinforceTrainer(_cfg(T=Tn, stop=stop), product).rollout(env, forced_actions=forced, start_positions=start)
    S = ref["rewards"].shape[1]
    assert ro["rewards"].shape[1] == S
    for k in ("masks", "logit_masks", "positions", "actions"):
        assert torch.equal(ro[k].cpu(), ref[k]), k
    assert torch.equal(ro["rewards"].cpu(), ref["rewards"])
    for k in ("returns", "logprobs", "entropies", "logits"):
        assert (ro[k].cpu() - ref[k]).abs().max() < TOL_LOGIT, k
    assert torch.equal(ro["patches"].cpu(), ref["patches"])       # bit exact gather inside the loop


def test_rollout_config5_sizes_vs_oracle(monkeypatch):
    """BASELINE configs[4] at its real sizes on a small batch: gpt-mini (6 layers, 6 heads, C = 192) + yolox-s dense-3x3
    encoder, 640-px patches (20 x 20 deepest map), forced rollout against the CPU oracle."""
    from oracle import env_ref, rollout_ref
    P, Tn, B = 640, 3, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, model_type="gpt-mini", gpt_backbone="yolox-s",
                                with_detector=False, image_processor=None, max_batch=B)
    images, bboxes, start = synth_batch(B, 2, 3, P, seed=77)
    forced = torch.tensor([[1, 3, 0], [2, 0, 3]])
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), forced_actions=forced,
                                  start_positions=start, stop_early=True)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    ro = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, forced_actions=forced, start_positions=start)
    for k in ("masks", "logit_masks", "positions", "actions"):
        assert torch.equal(ro[k].cpu(), ref[k]), k
    assert torch.equal(ro["rewards"].cpu(), ref["rewards"])
    for k in ("returns", "logprobs", "entropies", "logits"):
        assert (ro[k].cpu() - ref[k]).abs().max() < 1e-3, k              # north-star bound
    # the opt-in MFMA step kernel (kernels_gptmfma.hip) walks the same trajectory: env-step mode of that kernel
    for agents in ("4", "16"):
        monkeypatch.setenv("JN_GPT_MFMA", "1")
        monkeypatch.setenv("JN_GPT_MFMA_AGENTS", agents)
        env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
        rm = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, forced_actions=forced, start_positions=start)
        for k in ("masks", "positions", "actions", "rewards"):
            assert torch.equal(rm[k], ro[k]), k
        for k in ("logprobs", "entropies", "logits"):
            assert (rm[k] - ro[k]).abs().max() < 1e-5, k


def test_rollout_early_stop_and_full_length():
    """All agents STOP at step 2: reference breaks (reinforce.py:181-184) -> S == 2."""
    from oracle import env_ref, rollout_ref
    P, Tn, B = 64, 6, 3
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, image_processor="yolox-nano")
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=2)
    forced = torch.tensor([[1, 8, 0, 0, 0, 0], [8, 3, 0, 0, 0, 0], [2, 8, 0, 0, 0, 0]])
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    ro = tr.rollout(env, forced_actions=forced, start_positions=start)
    assert ro["rewards"].shape == (B, 2) and ro["masks"].shape == (B, 3) and ro["patches"].shape[1] == 3
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), forced_actions=forced,
                                  start_positions=start)
    assert torch.equal(ro["masks"].cpu(), ref["masks"]) and torch.equal(ro["rewards"].cpu(), ref["rewards"])
    assert (ro["returns"].cpu() - ref["returns"]).abs().max() < 1e-6
    full = tr.rollout(env, forced_actions=forced, start_positions=start, stop_early=False)
    assert full["rewards"].shape == (B, Tn)
    assert torch.equal(full["rewards"][:, :2].cpu(), ref["rewards"])


def test_rollout_sampling_statistics():
    """Sampled mode cannot match torch's RNG stream; check the action histogram of step 0
    against the softmax of the (deterministic) step-0 logits with a chi-square bound."""
    P, Tn, B = 32, 2, 1
    product, _ = make_pair(5, patch_size=P, block_size=Tn, image_processor="yolox-nano", max_batch=1)
    with torch.no_grad():
        product.action_head.lm_heads._modules["0"].weight.mul_(12.0)     # spread the distribution
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=4)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    n, counts, probs = 600, torch.zeros(9), None
    for i in range(n):
        ro = tr.rollout(env, sample_actions=True, start_positions=start, keep_patches=False, stop_early=False)
        counts[int(ro["actions"][0, 0])] += 1
        if probs is None:
            probs = torch.softmax(ro["logits"][0, 0].cpu(), -1)
            assert torch.isclose(ro["logprobs"][0, 0].cpu(), torch.log(probs[int(ro["actions"][0, 0])]), atol=1e-5)
            assert torch.isclose(ro["entropies"][0, 0].cpu(), -(probs * probs.log()).sum(), atol=1e-5)
    big = n * probs >= 5                                           # lump rare actions into one bin
    obs = torch.cat((counts[big], counts[~big].sum()[None]))
    exp = torch.cat((n * probs[big], n * probs[~big].sum()[None])).clamp(min=1e-9)
    chi2 = float((((obs - exp) ** 2) / exp).sum())
    assert chi2 < 32.9, (chi2, counts.tolist(), probs.tolist())    # chi2(<=8 dof) 99.99th percentile
    assert probs.max() < 0.9 and int(big.sum()) >= 3


def test_sampled_rollouts_at_the_headline_batch_follow_the_policy():
    """Sampled mode at the headline batch (B = 64 agents): the device sampler cannot reproduce
    torch's RNG stream, so the `--sample` line stands on statistics — over R rollouts every executed (agent,
    step) draws its action from softmax(logits of that step) (src/reinforce.py:73-90).  Pooled over agents, steps and runs:
    for each of the 9 actions the number of times it was drawn against the sum of its probabilities (a martingale:
    variance = sum p (1 - p)), a chi-square over the 9 pooled cells, the same for step 0 alone (start positions fixed: its
    logits repeat across runs) and the empirical STOP rate against the mean STOP probability; equal seeds repeat the
    trajectories, different seeds do not; agents draw independent streams."""
    P, Tn, B, R = 64, 5, 64, 24
    product, _ = make_pair(5, patch_size=P, block_size=Tn, image_processor="yolox-nano", max_batch=B)
    with torch.no_grad():
        product.action_head.lm_heads._modules["0"].weight.mul_(10.0)     # spread the distribution over the agents
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=14)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    obs, exp, var = torch.zeros(9, dtype=torch.float64), torch.zeros(9, dtype=torch.float64), torch.zeros(9, dtype=torch.float64)
    obs0, exp0, var0 = torch.zeros_like(obs), torch.zeros_like(obs), torch.zeros_like(obs)
    first_actions, n_valid = [], 0
    for r in range(R):
        ro = tr.rollout(env, sample_actions=True, start_positions=start, keep_patches=False)
        S = ro["actions"].shape[1]
        valid = ro["logit_masks"].cpu()                                # steps an agent actually took (before / at its STOP)
        pr = torch.softmax(ro["logits"].cpu().double(), -1)            # [B, S, 9]
        act = ro["actions"].cpu()
        onehot = torch.nn.functional.one_hot(act, 9).double()
        w = valid[..., None].double()
        obs += (onehot * w).sum((0, 1)); exp += (pr * w).sum((0, 1)); var += (pr * (1 - pr) * w).sum((0, 1))
        obs0 += onehot[:, 0].sum(0); exp0 += pr[:, 0].sum(0); var0 += (pr[:, 0] * (1 - pr[:, 0])).sum(0)
        n_valid += int(valid.sum())
        first_actions.append(act[:, 0].clone())
        lp = torch.log_softmax(ro["logits"].cpu().double(), -1).gather(-1, act[..., None])[..., 0]
        assert ((ro["logprobs"].cpu().double() - lp).abs() * valid).max() < 1e-5
        ent = -(pr * pr.clamp_min(1e-300).log()).sum(-1)
        assert ((ro["entropies"].cpu().double() - ent).abs() * valid).max() < 1e-5
        if r == 0:
            assert S >= 2 and float(pr[:, 0].max()) < 0.98                 # a real distribution, not an arg-max in disguise
    z = (obs - exp) / var.sqrt()
    z0 = (obs0 - exp0) / var0.sqrt()
    assert n_valid > 2000 and float(z.abs().max()) < 4.5, (z.tolist(), obs.tolist(), exp.tolist())
    assert float(z0.abs().max()) < 4.5, (z0.tolist(), obs0.tolist(), exp0.tolist())
    chi2 = float(((obs - exp) ** 2 / exp.clamp_min(1e-9)).sum())
    assert chi2 < 45.0, (chi2, obs.tolist(), exp.tolist())             # chi2(8 dof): 1e-6 tail at 42.7
    stop_rate, stop_p = float(obs0[8]) / (R * B), float(exp0[8]) / (R * B)
    assert abs(stop_rate - stop_p) < 4.5 * float(var0[8].sqrt()) / (R * B) and stop_p > 0.01, (stop_rate, stop_p)
    fa = torch.stack(first_actions)                                   # [R, B]
    assert (fa != fa[0]).any(dim=0).float().mean() > 0.5               # runs differ (new seed per rollout) ...
    assert len({tuple(fa[:, b].tolist()) for b in range(B)}) > B // 2  # ... and so do the agents' streams
    tr_a, tr_b = ja.ReinforceTrainer(_cfg(T=Tn), product), ja.ReinforceTrainer(_cfg(T=Tn), product)
    ra = tr_a.rollout(env, sample_actions=True, start_positions=start, keep_patches=False)
    rb = tr_b.rollout(env, sample_actions=True, start_positions=start, keep_patches=False)
    assert torch.equal(ra["actions"], rb["actions"]) and torch.equal(ra["positions"], rb["positions"])     # same seed, same walk


def test_config5_rollout_at_its_full_sequence_length():
    """configs[4] at ITS sequence length: gpt-mini (6 layers / 6 heads / C = 192) + yolox-s
    encoder, 640 px, T = 32 — the 33-token KV cache of src/models/gpt.py:481-534 had only been exercised to T = 3.  Forced
    non-STOP rollout of 2 agents over all 32 steps: integer state against the oracle's environment replay (positions,
    rewards, masks), returns = suffix sums, log-prob / entropy against the logits, and an ORACLE spot check of the first and
    the last steps (t = 0, 1, 30, 31): the oracle embeds step t's patch and runs the transformer over the prefix of the
    engine's own token embeddings (``prev_embeddings``, the reference's recurrent call) — logits of step t and the new
    token's embedding within 1e-4 / 2e-4, i.e. attention over the full cached prefix with gpt-mini's head size 32."""
    from oracle import env_ref
    P, Tn, B = 640, 32, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, model_type="gpt-mini", gpt_backbone="yolox-s",
                                with_detector=False, image_processor=None, max_batch=B)
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=47)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(9))
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    ro = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, forced_actions=forced, start_positions=start)
    assert ro["rewards"].shape == (B, Tn) and ro["final_emb"].shape == (B, Tn + 1, 192) and ro["patches"].shape[1] == Tn + 1
    # integer state: the oracle's environment replayed with the same actions (no model involved)
    eo = env_ref.EnvRef(images, bboxes, P, Tn, 1, True)
    eo.reset(start)
    pos, rew, term = [eo.positions.clone()], [], []
    for t in range(Tn):
        _, r, terminated, truncated, infos = eo.step(forced[:, t])
        pos.append(infos["positions"].clone()); rew.append(r); term.append(terminated)
    assert torch.equal(ro["positions"].cpu(), torch.stack(pos, 1))
    assert torch.equal(ro["rewards"].cpu(), torch.stack(rew, 1))
    assert bool(ro["masks"].all()) and bool(ro["logit_masks"].all()) and not bool(torch.stack(term, 1).any())
    rewards = ro["rewards"].cpu().double()
    suffix = rewards.flip(1).cumsum(1).flip(1)
    assert (ro["returns"].cpu().double() - suffix).abs().max() < 1e-5
    lg = ro["logits"].cpu().double()
    lp = torch.log_softmax(lg, -1)
    assert (ro["logprobs"].cpu().double() - lp.gather(-1, forced[..., None])[..., 0]).abs().max() < 1e-5
    assert (ro["entropies"].cpu().double() + (lp.exp() * lp).sum(-1)).abs().max() < 1e-5
    assert torch.equal(ro["actions"].cpu(), forced)
    # oracle spot check of single steps on the engine's own prefix
    patches, positions, emb = ro["patches"].cpu(), ro["positions"].cpu(), ro["final_emb"].cpu()
    tokens = torch.cat((torch.zeros((B, 1), dtype=torch.long), forced[:, :-1]), 1)       # action token of step t = action t-1 (BOS 0)
    classes = torch.zeros(B, dtype=torch.long)
    oracle.eval()
    for t in (0, 1, Tn - 2, Tn - 1):
        prev = None if t == 0 else emb[:, :t + 1]
        with torch.no_grad():
            lo, eo_ = oracle(patches[:, :t + 1], tokens[:, :t + 1], classes, positions[:, :t + 1], prev)
        assert eo_.shape == (B, t + 2, 192)
        assert (lo[:, -1] - ro["logits"][:, t].cpu()).abs().max() < TOL_LOGIT, t
        assert (eo_[:, -1] - emb[:, t + 1]).abs().max() < 2e-4, t
        if t == 0:
            assert (eo_[:, 0] - emb[:, 0]).abs().max() < 1e-6          # the class token


def test_full_size_c3_properties():
    """BASELINE config 3 sizes (448 px, seq-len 20, STOP, 4480x4480 images) at a batch the
    test box holds comfortably; size-independent properties + an oracle spot check."""
    from oracle import env_ref
    P, Tn, B, G = 448, 20, 8, 10
    product, oracle = make_pair(1, patch_size=P, block_size=Tn, with_detector=False, image_processor=None,
                                max_batch=B)
    gen = torch.Generator(device=DEV).manual_seed(12345)
    images = torch.rand((B, 3, G * P, G * P), device=DEV, generator=gen)
    g = torch.Generator().manual_seed(12345)
    bboxes = torch.zeros((B, 3, 4), dtype=torch.long)
    for b in range(B):
        for k in range(int(torch.randint(1, 4, (1,), generator=g))):
            w, h = (int(torch.randint(32, P, (1,), generator=g)) for _ in range(2))
            x, y = int(torch.randint(0, G * P - w, (1,), generator=g)), int(torch.randint(0, G * P - h, (1,), generator=g))
            bboxes[b, k] = torch.tensor([x, y, x + w, y + h])
    start = torch.randint(0, G, (B, 2), generator=g)
    forced = torch.randint(0, 8, (B, Tn), generator=g)              # non-STOP: S == T
    env = ja.NeedleGeneralEnv(images, bboxes, P, Tn, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    ro = tr.rollout(env, forced_actions=forced, start_positions=start)
    assert ro["rewards"].shape == (B, Tn)
    # (1) env trajectory equals the oracle env replay (integer exact); images are not needed for it
    envo = env_ref.EnvRef((B, 3, G * P, G * P), bboxes, P, Tn, 1, True)
    _, info = envo.reset(start)
    pos, rew = [info["positions"].clone()], []
    for t in range(Tn):
        _, r, te, trc, info = envo.step(forced[:, t])
        rew.append(r); pos.append(info["positions"].clone())
    assert torch.equal(ro["positions"].cpu(), torch.stack(pos, 1))
    assert torch.equal(ro["rewards"].cpu(), torch.stack(rew, 1))
    # (2) patches are bit-exact slices of the images at the visited positions
    p = ro["positions"]
    for b in range(B):
        for t in (0, 7, Tn):
            y, x = p[b, t].tolist()
            assert torch.equal(ro["patches"][b, t], images[b, :, y * P:(y + 1) * P, x * P:(x + 1) * P])
    # (3) returns are suffix sums of the masked rewards; logit_masks is the rolled mask
    lm = ro["logit_masks"].float()
    sfx = torch.flip(torch.cumsum(torch.flip(ro["rewards"] * lm, (1,)), 1), (1,))
    assert torch.allclose(ro["returns"], sfx, atol=1e-5)
    assert torch.equal(ro["logit_masks"][:, 1:], ro["masks"][:, 1:-1]) and bool(ro["logit_masks"][:, 0].all())
    # (4) log-probs / entropies are consistent with the logits; probabilities sum to one
    lsm = torch.log_softmax(ro["logits"], -1)
    assert torch.allclose(lsm.gather(2, ro["actions"][..., None]).squeeze(2), ro["logprobs"], atol=1e-5)
    assert torch.allclose(-(lsm.exp() * lsm).sum(-1), ro["entropies"], atol=1e-5)
    # (5) oracle spot check on 2 agents x 3 steps at full patch size (logits within 1e-3, north star)
    with torch.no_grad():
        for b in (0, B - 1):
            e = None
            for t in range(3):
                pt = ro["patches"][b:b + 1, :t + 1].cpu()
                acts = torch.cat((torch.zeros(1, 1, dtype=torch.long), forced[b:b + 1, :t]), 1)
                lg, e = oracle(pt, acts, torch.zeros(1, dtype=torch.long), ro["positions"][b:b + 1, :t + 1].cpu(), e)
                assert (lg[0, -1] - ro["logits"][b, t].cpu()).abs().max() < 1e-3


# --------------------------------------------------------------------------------------
# training: one REINFORCE iteration (src/reinforce.py:302-353)
# --------------------------------------------------------------------------------------
GRAD_TOL = 1e-3


# Gradient bars (max|got - ref| <= bar * max|ref| per tensor; north star 1e-3):
#   * decision side (transformer, embeddings, heads, embed_fpn.3): 1e-3;
#   * patch-encoder tensors: 3e-3 — every one of them sits behind a chain of train-mode BatchNorms over the near-constant
#     maps of a random-init net, which amplifies fp32 rounding: across seeds and sizes the product lands between 1e-4 and
#     2.6e-3 of the fp32 CPU oracle, and the fp32 oracle itself between 1e-4 and 1.6e-3 of its own fp64 evaluation
#     (measured here: 1.6e-3 at 64 px, 3.6e-3 at 448 px); 5e-3 for the first three stages (stem, dark2, dark3: the end of
#     the 77-layer backward chain);
#   * at the full patch size (448 px) the tests also pass `ref64` (gradients of the fp64 oracle): the product is compared
#     with the fp64 values and a tensor may additionally use NOISE_FACTOR x the fp32 oracle's own distance from fp64, and
#     twice the stage bars above ("as accurate as fp32 arithmetic allows");
#   * full size again, `probe` (the conditioning probe, _conditioning_probe below): the fp32 oracle re-run with every
#     SiLU output perturbed by a relative 1e-7 (about one ulp — what ANY other fp32 evaluation order does).  Its distance
#     from fp64 is what the problem's conditioning allows: with B * T = 32 patches the deep maps hold ReLU / max-pool
#     near-ties that flip under such a perturbation (embed_fpn.0.weight then sits 1.35e-2 off, at 1e-7 and at 4e-7 alike;
#     the engine sits 1.38e-2 off whatever its rounding — native or ~1-ulp SiLU, exact or split-bf16 GEMM).  A tensor may
#     use PROBE_FACTOR x that distance, or FAMILY_FACTOR x the worst distance in its family (encoder / embed_fpn / rest).
#   * embed_fpn.0.weight at full size: 5e-3 (measured 2.2e-3 at 448 px, T = 20).
# JN_TEST_GRAD_REPORT=<file> appends the worst tensors of every call (a measuring aid).
#   * RELATIVE-L2 bars (round 3), beside the max-norm bars and WITHOUT any probe / noise relaxation, at every size:
#     ||got - ref||_2 <= L2_DECISION * ||ref||_2 for the decision side, L2_ENCODER for the patch encoder and embed_fpn.0
#     (ref = the fp64 oracle where the test computes one, else the fp32 oracle).  A handful of flipped ReLU / max-pool
#     near-ties moves single elements (the max-norm) but not the tensor: the tensors that need the probe allowance in
#     max-norm are listed with both distances in the JN_TEST_GRAD_REPORT file.
NOISE_FACTOR = 4.0
PROBE_FACTOR = 3.0
FAMILY_FACTOR = 2.0
#     Measured (round 3, gpurun_out -> profiles/r03_grad_report.txt): the fixed L2 bars hold everywhere (worst 1.1e-3 of 3e-3
#     at the REINFORCE 448 px / T = 20 and headline-mix cases) except in the two cases whose gradient passes through FEW
#     decision pixels: the supervised step at B * T = 32 patches of 448 px (product 3 - 4.3e-3 on a dozen deep-FPN tensors) and
#     configs[4] at 640 px with 4 patches (7e-3 - 1e-2 on ALL encoder tensors).  There the L2 distance is a continuous,
#     measurable function of the FORWARD rounding error: the fp32 CPU oracle itself, re-run with relative noise eps on its
#     SiLU outputs, sits at L2 = 2e-4 (eps = 0), 1.6 - 3.8e-3 (1e-7, one ulp), 5.6e-3 median / 7.1e-3 worst (1e-6) and 1.9e-2
#     (1e-5) of its fp64 evaluation at 640 px — distance ~ sqrt(eps): the number of flipped embed_fpn ReLU / SPP arg-max
#     near-ties grows with eps, and ONE flipped ReLU re-routes the gradient of one of only B * T * h * w = 1 600 decision
#     pixels, one flipped 13 x 13 arg-max that of most of a 14 x 14 channel map.  So those two tests also compute
#     `probe_l2` — the relative-L2 distance of the noisy fp32 oracle (one ulp, three draws, for the supervised step; 4e-6
#     for configs[4], whose dense 3x3 layers accumulate thousands of products per output in MFMA order) — and a tensor may
#     use L2_PROBE_FACTOR x the worst probe distance in its family; their forward is pinned separately (logits within 1e-4
#     of the fp64 oracle).  Everywhere else the fixed L2 bars stand alone.
L2_DECISION = 1e-3
L2_ENCODER = 3e-3
L2_PROBE_FACTOR = 1.5


def l2_bar(name):
    return L2_ENCODER if name.startswith(("gpt_backbone.", "yolox.", "embed_fpn.0")) else L2_DECISION
FIRST_STAGES = ("gpt_backbone.backbone.stem.", "gpt_backbone.backbone.dark2.", "gpt_backbone.backbone.dark3.")


def grad_bar(name, full_size=False):
    if name.startswith(FIRST_STAGES):
        bar = 5e-3
    elif name.startswith(("gpt_backbone.", "yolox.")):
        bar = 3e-3
    else:
        bar = GRAD_TOL
    if full_size:
        bar = max(2.0 * bar if bar > GRAD_TOL else bar, 5e-3 if name == "embed_fpn.0.weight" else 0.0)
    return bar


class _UlpSiLU(torch.nn.Module):
    def __init__(self, gen, eps=1e-7):
        super().__init__()
        self.gen, self.eps = gen, eps

    def forward(self, x):
        y = torch.nn.functional.silu(x)
        return y * (1.0 + self.eps * torch.randn(y.shape, generator=self.gen, dtype=y.dtype))


def _with_noisy_silu(oracle, seed, eps):
    """A copy of the oracle whose every SiLU output carries relative Gaussian noise `eps`."""
    import copy
    o = copy.deepcopy(oracle)
    gen = torch.Generator().manual_seed(seed)

    def swap(mod):
        for n, c in list(mod.named_children()):
            if isinstance(c, torch.nn.SiLU):
                setattr(mod, n, _UlpSiLU(gen, eps))
            else:
                swap(c)
    swap(o)
    return o


def _measured_probe_eps(product, oracle, patches, eps0=2e-6):
    """The noise level of the conditioning probe, MEASURED instead of picked: the eps at which the
    noisy fp32 oracle's FORWARD error equals the engine's.  Train-mode FPN maps (batch statistics) of `patches`, relative L2
    over the three levels, against the fp64 oracle: e_eng (the engine), e32 (the fp32 oracle, torch's own rounding) and e0
    (the fp32 oracle with SiLU noise eps0).  The injected noise acts linearly on the maps and adds to torch's rounding in
    quadrature: k = sqrt(e0^2 - e32^2) / eps0, eps = sqrt(max(e_eng^2 - e32^2, 0)) / k.  Floor: one ulp (1e-7), the probe
    of an engine that rounds no worse than torch.  A forward error beyond eps = 1e-5 is a forward bug, not conditioning."""
    import copy
    enc = lambda o: o.gpt_backbone if getattr(o, "gpt_backbone", None) is not None else o.yolox.backbone
    o64 = copy.deepcopy(oracle).double().train()
    with torch.no_grad():
        ref = [m.double() for m in enc(o64)(patches.double())]
        f32 = [m.double() for m in enc(copy.deepcopy(oracle).train())(patches)]
        f0 = [m.double() for m in enc(_with_noisy_silu(oracle, 77, eps0).train())(patches)]
    got = [m.cpu().double() for m in product.backbone_features(patches, train=True)]
    rel = lambda maps: float(sum((a - b).pow(2).sum() for a, b in zip(maps, ref)) / sum(b.pow(2).sum() for b in ref)) ** 0.5
    e_eng, e32, e0 = rel(got), rel(f32), rel(f0)
    k = max(e0 ** 2 - e32 ** 2, 1e-30) ** 0.5 / eps0
    eps = max(e_eng ** 2 - e32 ** 2, 0.0) ** 0.5 / k
    import os
    rep = os.environ.get("JN_TEST_GRAD_REPORT")
    if rep:
        with open(rep, "a") as f:
            f.write(f"# measured probe eps: engine maps {e_eng:.3e}, fp32 oracle {e32:.3e}, oracle + noise {eps0:g}: {e0:.3e} "
                    f"(relative L2 vs fp64) -> eps = {eps:.3e}\n")
    assert eps <= 1e-5, ("the engine's train-mode maps are further from fp64 than the noisy oracle at eps = 1e-5", e_eng, e32, e0)
    return max(eps, 1e-7)


def _conditioning_probe(oracle, run, ref64, samples=1, eps=1e-7, l2=False, both=False, first=0):
    """{tensor: relative distance from fp64} of the fp32 oracle with ~1-ulp noise (`eps`) on every SiLU output (see the bars
    above; the worst of `samples` noise draws — whether a given near-tie flips is a matter of chance).
    `run(model)` performs forward + backward on the model it is given.  l2=True: relative-L2 distances instead of max-norm;
    both=True: (max-norm dict, relative-L2 dict) of the same draws."""
    out, out2 = {}, {}
    for k in range(first, first + samples):
        o = _with_noisy_silu(oracle, 1234 + k, eps)
        run(o)
        for n, p in o.named_parameters():
            if p.grad is not None and n in ref64:
                d2 = (p.grad.double() - ref64[n]).norm().item() / max(ref64[n].norm().item(), 1e-30)
                dm = (p.grad.double() - ref64[n]).abs().max().item() / max(ref64[n].abs().max().item(), 1e-30)
                out[n] = max(out.get(n, 0.0), d2 if (l2 and not both) else dm)
                out2[n] = max(out2.get(n, 0.0), d2)
    return (out, out2) if both else out


def _check_grads(grads, oracle, skip_prefix=("yolox",), tag="", ref64=None, probe=None, full_size=None, probe_l2=None):
    import os
    if full_size is None:
        full_size = ref64 is not None
    rows, checked = [], 0
    for name, p in oracle.named_parameters():
        if any(name.startswith(sp) for sp in skip_prefix) or p.grad is None or not p.requires_grad:
            continue
        gp, ref = grads[name].double(), p.grad.double()
        noise = 0.0
        if ref64 is not None:
            r64 = ref64[name]
            noise = (ref - r64).abs().max().item() / max(r64.abs().max().item(), 1e-30)
            ref = r64
        scale = ref.abs().max().item()
        if scale < 1e-12:
            assert gp.abs().max().item() < 1e-9, name
            continue
        err = (gp - ref).abs().max().item() / scale
        l2 = (gp - ref).norm().item() / max(ref.norm().item(), 1e-30)
        rows.append((err, name, scale, noise, l2))
        checked += 1
    rep = os.environ.get("JN_TEST_GRAD_REPORT")
    if rep:
        with open(rep, "a") as f:
            f.write(f"# {tag}: worst max-norm tensors (tensor, max-norm distance, max|ref|, fp32-oracle noise, relative L2, "
                    f"fixed max-norm bar, L2 bar)\n")
            for err, name, scale, noise, l2 in sorted(rows, reverse=True)[:25]:
                f.write(f"{tag}\t{name}\t{err:.3e}\t{scale:.3e}\t{noise:.3e}\tL2 {l2:.3e}\tbar {grad_bar(name, full_size):.1e}\t"
                        f"L2bar {l2_bar(name):.1e}{'  NEEDS-ALLOWANCE' if err >= grad_bar(name, full_size) else ''}\n")
            worst = max(rows, key=lambda r: r[4] / l2_bar(r[1]))
            f.write(f"# {tag}: worst relative L2 / bar: {worst[1]} {worst[4]:.3e} (bar {l2_bar(worst[1]):.1e})\n")
    def family(name):
        return next((f for f in ("gpt_backbone.", "yolox.", "embed_fpn.") if name.startswith(f)), "decision")
    fam_probe = {}
    for name, d in (probe or {}).items():
        fam_probe[family(name)] = max(fam_probe.get(family(name), 0.0), d)
    fam_probe_l2 = {}
    for name, d in (probe_l2 or {}).items():
        fam_probe_l2[family(name)] = max(fam_probe_l2.get(family(name), 0.0), d)
    if rep and probe_l2:
        with open(rep, "a") as f:
            f.write(f"# {tag}: family-worst L2 distance of the noisy fp32 oracle from fp64: {fam_probe_l2}\n")
    for err, name, scale, noise, l2 in rows:
        # (one probe run samples the near-ties once: a flip it shows on one tensor of a family can land on a sibling
        #  under other rounding — hence also FAMILY_FACTOR x the family's worst probe distance)
        bar = max(grad_bar(name, full_size), NOISE_FACTOR * noise, PROBE_FACTOR * (probe or {}).get(name, 0.0),
                  FAMILY_FACTOR * fam_probe.get(family(name), 0.0))
        assert err < bar, (fam_probe, tag, name, err, scale, noise, (probe or {}).get(name), sorted(rows, reverse=True)[:5])
        bar2 = max(l2_bar(name), L2_PROBE_FACTOR * fam_probe_l2.get(family(name), 0.0))
        assert l2 < bar2, ("relative L2", tag, name, l2, bar2, err, fam_probe_l2)
    return checked


def _check_grads_probed(grads, oracle, run, ref64, eps, tag, max_draws=16, **kw):
    """_check_grads with the conditioning probe drawn ON DEMAND (round 4).  What the CPU experiments of this round showed
    (scratch of the round): for a fixed input the encoder gradients of a train-mode REINFORCE / supervised step
    sit in one of a few discrete STATES — evaluations of the same algorithm that differ by one ulp somewhere in the forward
    (torch fp32, fp64, BatchNorm as fma(z, scale, shift) or centred, a 1-ulp SiLU) agree to <= 2e-3 relative L2 with each
    other inside a state and differ by 1.6e-2 across the two states of the headline-mix input, always on the same tensors
    (one near-tie of a max-pool arg-max / ReLU that resolves either way).  The engine's forward is as accurate as torch's
    (_measured_probe_eps: 2.1e-5 of fp64 on the maps, both) and lands in the other state than fp64 there.  So: the fixed
    bars first; only if they do not hold, one-ulp (eps, measured) perturbed evaluations of the oracle are drawn one at a
    time — seeds fixed, at most `max_draws` — and a tensor may use the probe allowance of _check_grads over the draws so
    far.  The draw budget is fixed up front; the maximum over more draws only grows, so stopping at the first pass is the
    same criterion as using all of them.  Returns (tensors checked, draws used)."""
    probe, probe_l2 = {}, {}
    for k in range(max_draws + 1):
        try:
            n = _check_grads(grads, oracle, tag=f"{tag} [probe eps {eps:.1e}, {k} draws]", ref64=ref64, probe=probe or None,
                             probe_l2=probe_l2 or None, **kw)
            return n, k
        except AssertionError:
            if k == max_draws:
                raise
        p, p2 = _conditioning_probe(oracle, run, ref64, samples=1, eps=eps, both=True, first=k)
        for name, d in p.items():
            probe[name] = max(probe.get(name, 0.0), d)
        for name, d in p2.items():
            probe_l2[name] = max(probe_l2.get(name, 0.0), d)


def _grads64(oracle):
    return {n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None}


def _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, stop, mean, std, ew):
    from oracle import env_ref, rollout_ref
    oracle.train()
    oracle.zero_grad()
    env = env_ref.EnvRef(images, bboxes, P, Tn, 1, stop)
    ro = rollout_ref.rollout(oracle, env, forced_actions=forced, start_positions=start)
    norm = rollout_ref.ReturnNormaliser()
    norm.mean, norm.std = mean, std
    m = rollout_ref.reinforce_metrics(ro, ew, norm)
    m["loss"].backward()
    return ro, m


@pytest.mark.parametrize("stop,B,P,Tn,grad_slots,arch", [
    (True, 3, 64, 4, None, {}), (False, 2, 96, 3, None, {}), (True, 3, 64, 4, 3, {}),
    (True, 3, 96, 3, None, dict(model_type="gpt-mini", gpt_backbone="yolox-s")),      # BASELINE config 5 topology
    (True, 2, 448, 20, None, {})])                                                     # BASELINE configs[2] patch / sequence sizes
def test_reinforce_iteration_gradients_vs_oracle(stop, B, P, Tn, grad_slots, arch, monkeypatch):
    """loss.backward() of a whole REINFORCE iteration (train-mode BN per glimpse step, T backbone
    passes, causal GPT over the trajectory) against torch autograd on the CPU oracle.  The backward is
    step-batched: all T passes per launch, or chunks of `grad_slots` passes when memory is capped."""
    if grad_slots:
        monkeypatch.setenv("JN_GRAD_SLOTS", str(grad_slots))
    nA = 9 if stop else 8
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, nclasses=nA, with_detector=False, image_processor=None, **arch)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=41)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
    # (448 px / T = 20: one fp32 oracle pass — three minutes of CPU time went into the fp64 and probe passes of this case
    #  alone; the full-size bars (twice the stage bars) apply, the fp64 / conditioning-probe treatment stays with the
    #  supervised step at 448 px, the worse-conditioned and much cheaper case)
    ro, m = _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, stop, 0.25, 1.5, 0.01)
    cfg = _cfg(T=Tn, stop=stop, learning_rate=1e-3, gradient_accumulation=1)
    tr = ja.ReinforceTrainer(cfg, product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, stop)
    got_m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    for k in ("action_loss", "entropy_loss", "loss", "returns", "episode_length"):
        assert abs(float(got_m[k]) - float(m[k].detach())) < 2e-4, (k, float(got_m[k]), float(m[k].detach()))
    checked = _check_grads(product.engine_grads(), oracle, tag=f"reinforce P={P} T={Tn} {arch.get('gpt_backbone', 'nano')}", full_size=P >= 448)
    assert checked > 150
    # running statistics moved Tn times, as in the reference's train-mode rollout
    product.pull_bn_statistics()
    k = "gpt_backbone.backbone.dark2.0.pconv.bn.running_mean" if not arch else "gpt_backbone.backbone.dark2.0.bn.running_mean"
    assert torch.allclose(product.state_dict()[k], oracle.state_dict()[k], atol=1e-5, rtol=1e-3)


def test_reinforce_iteration_at_the_headline_kernel_mix_vs_oracle():
    """The kernel-path mix of the headline batch (B = 64 at 448 px), which depends on pixels per launch: with
    21 <= B <= 83 the 56x56 (and larger) layers finalize their BatchNorm tables per layer while the 28x28 / 14x14 ones are
    deferred to their consumers (JN_DEFER_MAX_M = 65536 pixels), so C3_p3.conv2|conv1 reads a concat whose upsampled half
    has deferred entries and whose dark3 half has not, bu_conv2.dconv crosses the boundary, etc.; the narrow persistent
    1x1 kernel, the weight-stationary wide one and the fused backward kernels run with the grids of thousands of
    pixel tiles.  B = 24, 448 px, T = 2: train-mode maps of the first glimpse step, losses and every gradient of the
    REINFORCE iteration against torch autograd on the CPU oracle (src/reinforce.py:302-353, src/models/gpt.py:356-384)."""
    P, Tn, B = 448, 2, 24
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, max_batch=B)
    images, bboxes, start = synth_batch(B, 2, 2, P, seed=43)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
    assert 21 * 56 * 56 > 65536 >= B * 28 * 28            # the boundary sits where the headline batch has it
    # 48 patches: every gradient passes through 48 * 196 decision pixels and 13 x 13 arg-max windows — the regime in which ONE
    # flipped near-tie moves whole tensors (see the bars above).  Measured on the CPU oracle for this input: fp32 vs fp64
    # median 9.5e-4 / worst 1.9e-3 relative L2; with one-ulp SiLU noise 2.2e-3 / 4.7e-3; with 1e-6 7.8e-3 / 1.1e-2.  The
    # engine was at 1.1e-3 in three runs of round 3 and at 1.6e-2 after a change that only re-grouped fp32 partial sums (two
    # forward evaluations that agree to 1e-5 on the maps).  So: fp64 reference, the probe (both norms) at the noise level
    # that reproduces the engine's MEASURED forward error on the start patches (_measured_probe_eps; round 3 used a constant,
    # 2e-6), forward pinned separately (logits within 1e-4 of fp64); a kernel-path bug at these launch shapes is O(1).
    import copy
    run = lambda o, dt=torch.float32: _oracle_reinforce_grads(o, images.to(dt), bboxes, start, forced, P, Tn, True, 0.25, 1.5, 0.01)
    o64 = copy.deepcopy(oracle).double()
    ro64, _ = run(o64, torch.float64)
    ref64 = _grads64(o64)
    ro, m = run(oracle)
    tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    got_m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    grads = product.engine_grads()
    assert (tr._last_train_buffers["logits"].cpu().double() - ro64["logits"].detach()).abs().max() < 1e-4
    for k in ("action_loss", "entropy_loss", "loss", "returns", "episode_length"):
        assert abs(float(got_m[k]) - float(m[k].detach())) < 2e-4, (k, float(got_m[k]), float(m[k].detach()))
    y0, x0 = start[:, 0], start[:, 1]
    patches0 = torch.stack([images[b, :, y0[b] * P:(y0[b] + 1) * P, x0[b] * P:(x0[b] + 1) * P] for b in range(B)])
    eps = _measured_probe_eps(product, oracle, patches0)
    n_checked, draws = _check_grads_probed(grads, oracle, run, ref64, eps, tag=f"reinforce headline mix B={B} P={P} T={Tn}")
    assert n_checked > 150
    # train-mode maps of the start patches (batch statistics over the 24 patches), all three FPN levels
    oracle.train()
    with torch.no_grad():
        want = oracle.gpt_backbone(patches0)
    got = product.backbone_features(patches0, train=True)
    for g_, w_ in zip(got, want):
        assert (g_.cpu() - w_).abs().max().item() < 1e-3 * max(1.0, w_.abs().max().item())


def test_train_mode_backbone_pass_at_the_headline_batch_vs_oracle():
    """ONE train-mode pass of the patch encoder over 64 patches of 448 px — the launch shapes bench.py times (M = 50 176
    pixels on the 28x28 maps, 12 544 on the 14x14 ones, per-layer finalize above 56x56) — against the CPU oracle: the
    three FPN maps within 1e-3 and the running statistics of a late layer (batch statistics over the 64 patches)."""
    N, P = 64, 448
    product, oracle = make_pair(3, patch_size=P, block_size=2, with_detector=False, image_processor=None, max_batch=N)
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(11))
    oracle.train()
    with torch.no_grad():
        want = oracle.gpt_backbone(x)
    got = product.backbone_features(x, train=True)
    for g_, w_ in zip(got, want):
        assert (g_.cpu() - w_).abs().max().item() < 1e-3 * max(1.0, w_.abs().max().item())
    product.pull_bn_statistics()
    for k in ("gpt_backbone.C3_n4.conv3.bn.running_var", "gpt_backbone.backbone.dark4.1.conv3.bn.running_mean",
              "gpt_backbone.backbone.dark2.0.pconv.bn.running_mean"):
        assert torch.allclose(product.state_dict()[k].cpu(), oracle.state_dict()[k], atol=1e-5, rtol=1e-3), k


def test_three_way_split_1x1_kernels_are_as_accurate_as_the_fp32_matrix_pipe(monkeypatch):
    """Round 3: the small-map 1x1 layers run on the bf16 matrix pipe with every fp32 operand split into three bf16 values
    (pw_x3_kernel: six products per pair, the three below 2^-25 dropped).  The claim is fp32 accuracy: a train-mode pass
    over 16 patches of 224 px (28x28 ... 7x7 maps, K, N = 64 ... 256 on that route) with the route on and off
    (JN_NO_PW_X3=1: v_mfma_f32_16x16x4_f32) against the oracle in fp64 — the split route must not be further from fp64
    than the fp32 pipe is (x 1.5 + one part in 1e6), and it must really have been taken (outputs differ)."""
    N, P = 16, 224
    product, oracle = make_pair(3, patch_size=P, block_size=2, with_detector=False, image_processor=None, max_batch=N)
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(17))
    oracle.train()
    o64 = copy.deepcopy(oracle).double()
    with torch.no_grad():
        want = o64.gpt_backbone(x.double())
    on = [t.cpu().double() for t in product.backbone_features(x, train=True)]
    monkeypatch.setenv("JN_NO_PW_X3", "1")
    off = [t.cpu().double() for t in product.backbone_features(x, train=True)]
    monkeypatch.delenv("JN_NO_PW_X3")
    assert any(not torch.equal(a, b) for a, b in zip(on, off))
    for a, b, w in zip(on, off, want):
        scale = w.abs().max().item()
        e_on, e_off = (a - w).abs().max().item() / scale, (b - w).abs().max().item() / scale
        assert e_off < 1e-4 and e_on < 1.5 * e_off + 1e-6, (e_on, e_off)


def test_headline_shape_backward_does_not_depend_on_the_step_batching(monkeypatch):
    """BASELINE configs[2] shape (64 agents, 448 px, T = 20): the step-batched backward (ONE set of launches over all 20
    glimpse steps: grid.z = 20, the 'whole resident rounds' grid rules) against the same iteration differentiated in chunks
    of 5 steps (JN_GRAD_SLOTS; grids of the sizes the oracle tests cover).  A size-independent property: the loss and every
    gradient agree to the order of the atomics (relative L2 <= 2e-5, max-norm <= 2e-4)."""
    P, Tn, B, G = 448, 20, 64, 3
    images = torch.rand((B, 3, G * P, G * P), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    _, bboxes, start = synth_batch(B, G, G, 64, seed=41)
    bboxes = bboxes * (P // 64)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
    grads, losses = [], []
    for slots in (None, 5):
        if slots:
            monkeypatch.setenv("JN_GRAD_SLOTS", str(slots))
        product, _ = make_pair(5, bn_seed=None, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, max_batch=B)
        tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
        env = ja.NeedleGeneralEnv(images, bboxes, P, Tn, 1, True)
        m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
        assert m["steps"] == Tn
        losses.append(float(m["loss"]))
        grads.append(product.engine_grads())
        del product, tr, env
        torch.cuda.empty_cache()
    assert abs(losses[0] - losses[1]) < 1e-5
    checked = 0
    for k, a in grads[0].items():
        b = grads[1][k]
        if float(a.abs().max()) < 1e-12:
            continue
        assert float((a - b).norm() / a.norm()) < 2e-5, (k, float((a - b).norm() / a.norm()))
        assert float((a - b).abs().max() / a.abs().max()) < 2e-4, k
        checked += 1
    assert checked > 150


def test_config5_training_at_its_sequence_length_does_not_depend_on_the_step_batching(monkeypatch):
    """configs[4] TRAINING at its sequence length (gpt-mini + yolox-s dense-3x3 encoder, 640 px, T = 32; B = 2): the oracle's
    autograd over 64 patches of 640 px is minutes of CPU time, so the full length is covered by a size-independent property
    of the engine — the step-batched backward over all 32 glimpse steps (grid.z = 32, the 33-token teacher-forced GPT
    backward) against the same iteration differentiated in chunks of 8 steps: loss and every gradient agree to the order of
    the atomics.  (Values at this topology are checked against the oracle at T = 2, rollouts at T = 32: the tests around.)"""
    P, Tn, B, G = 640, 32, 2, 3
    images = torch.rand((B, 3, G * P, G * P), device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    _, bboxes, start = synth_batch(B, G, G, 64, seed=43)
    bboxes = bboxes * (P // 64)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(4))
    grads, losses = [], []
    for slots in (None, 8):
        if slots:
            monkeypatch.setenv("JN_GRAD_SLOTS", str(slots))
        product, _ = make_pair(5, bn_seed=None, patch_size=P, block_size=Tn, model_type="gpt-mini", gpt_backbone="yolox-s",
                               with_detector=False, image_processor=None, max_batch=B)
        tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
        env = ja.NeedleGeneralEnv(images, bboxes, P, Tn, 1, True)
        m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
        assert m["steps"] == Tn and np.isfinite(float(m["loss"]))
        losses.append(float(m["loss"]))
        grads.append(product.engine_grads())
        del product, tr, env
        torch.cuda.empty_cache()
    assert abs(losses[0] - losses[1]) < 1e-5
    checked = 0
    for k, a in grads[0].items():
        b = grads[1][k]
        if float(a.abs().max()) < 1e-12:
            continue
        assert float((a - b).norm() / a.norm()) < 5e-5, (k, float((a - b).norm() / a.norm()))
        assert float((a - b).abs().max() / a.abs().max()) < 5e-4, k
        checked += 1
    assert checked > 150


def test_config5_training_at_its_patch_size_vs_oracle():
    """BASELINE configs[4] topology at its REAL patch size: gpt-mini + yolox-s (dense 3x3) encoder, 640 px — B = 2, T = 2
    REINFORCE iteration (train-mode BatchNorm per glimpse step), logits, losses and every gradient against torch autograd
    on the CPU oracle (src/reinforce.py:302-353; src/models/gpt.py:77-108, 137-140, 212-214)."""
    P, Tn, B = 640, 2, 2
    arch = dict(model_type="gpt-mini", gpt_backbone="yolox-s")
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, max_batch=B, **arch)
    images, bboxes, start = synth_batch(B, 2, 2, P, seed=45)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
    import copy
    run = lambda o, dt=torch.float32: _oracle_reinforce_grads(o, images.to(dt), bboxes, start, forced, P, Tn, True, 0.25, 1.5, 0.01)
    o64 = copy.deepcopy(oracle).double()
    ro64, _ = run(o64, torch.float64)
    ref64 = _grads64(o64)
    # 4 patches: every gradient passes through 1 600 decision pixels — the distance follows the forward rounding error
    # (see the bars above; measured on the CPU oracle at this size: L2 5.6e-3 at eps = 1e-6, 1.9e-2 at 1e-5, ~ sqrt(eps)).
    # The dense 3x3 layers of yolox-s accumulate up to 4 608 products per output in fp32 MFMA order: the engine lands where
    # the oracle lands with eps ~ 3e-6 (its logits stay within 1e-4 of fp64, asserted below; the oracle's move by 4.5e-5 at
    # eps = 1e-6 and 4.4e-4 at 1e-5).  Round 3 probed at a constant 4e-6; the probe now runs at the eps that reproduces the
    # engine's MEASURED forward error on the start patches (_measured_probe_eps), both norms.
    ro, m = run(oracle)
    tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    got_m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    grads = product.engine_grads()
    assert (tr._last_train_buffers["logits"].cpu().double() - ro64["logits"].detach()).abs().max() < 1e-4     # the forward, pinned
    for k in ("action_loss", "entropy_loss", "loss", "returns", "episode_length"):
        assert abs(float(got_m[k]) - float(m[k].detach())) < 2e-4, (k, float(got_m[k]), float(m[k].detach()))
    y0, x0 = start[:, 0], start[:, 1]
    patches0 = torch.stack([images[b, :, y0[b] * P:(y0[b] + 1) * P, x0[b] * P:(x0[b] + 1) * P] for b in range(B)])
    eps = _measured_probe_eps(product, oracle, patches0)
    n_checked, draws = _check_grads_probed(grads, oracle, run, ref64, eps, tag=f"reinforce c5 P={P} T={Tn} gpt-mini + yolox-s", max_draws=16)
    assert n_checked > 150


@pytest.mark.parametrize("mode", ["reinforce", "supervised"])
def test_dropout_training_with_injected_masks_vs_oracle(mode):
    """--dropout 0.1 (the reference's default, main.py:123-128): embd / attn / resid dropout of the decision transformer
    in the train-mode passes.  The engine's keep masks are a pure function of (seed, agent, token, layer, site, index)
    (Philox4x32-10, regenerated in the backward); the oracle applies the same masks (oracle/dropout_ref.py) at the
    reference's four sites: logits, loss and every gradient agree; eval-mode passes do not drop."""
    P, Tn, B, pdrop, seed = 64, 4, 3, 0.1, 77
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, dropout=pdrop)
    assert product.dropout == pdrop
    oracle.enable_dropout(pdrop, seed)
    product.set_dropout_seed(seed)
    if mode == "reinforce":
        images, bboxes, start = synth_batch(B, 3, 4, P, seed=41)
        forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
        ro, m = _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, True, 0.25, 1.5, 0.01)
        tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
        tr.last_return_mean, tr.last_return_std = 0.25, 1.5
        env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
        got = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
        assert (tr._last_train_buffers["logits"].cpu() - ro["logits"].detach()).abs().max() < 1e-3
        for k in ("action_loss", "entropy_loss", "loss"):
            assert abs(float(got[k]) - float(m[k])) < 2e-4, k
        # the masks matter: the same iteration without dropout gives other logits
        oracle.enable_dropout(0.0, 0)
        ro0, _ = _oracle_reinforce_grads(build_like(oracle), images, bboxes, start, forced, P, Tn, True, 0.25, 1.5, 0.01)
        assert (ro0["logits"] - ro["logits"]).abs().max() > 1e-3
        # eval-mode rollout: no dropout
        product.eval()
        ev = tr.rollout(ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True), forced_actions=forced, start_positions=start)
        oracle.eval()
        from oracle import env_ref, rollout_ref
        with torch.no_grad():
            ref_ev = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), forced_actions=forced,
                                         start_positions=start)
        assert (ev["logits"].cpu() - ref_ev["logits"]).abs().max() < 1e-3
    else:
        patches, cur, positions = synth_tokens(B, Tn, P, 9, 5, seed=21)
        nxt = torch.randint(0, 9, (B, Tn), generator=torch.Generator().manual_seed(4))
        masks = torch.ones((B, Tn), dtype=torch.long)
        oracle.train()
        oracle.zero_grad()
        logits, _ = oracle(patches, cur, torch.zeros(B, dtype=torch.long), positions)
        torch.nn.functional.cross_entropy(logits.reshape(B * Tn, 9), nxt.flatten()).backward()
        product2, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, dropout=pdrop,
                                max_batch=B * Tn)
        product2.set_dropout_seed(seed)
        cfg = ja.CfgNode(stop_enabled=True, stop_weight=1.0, learning_rate=1e-3, gradient_accumulation=1)
        mm = ja.SupervisedTrainer(cfg, product2).train_step(patches, cur, nxt, positions, masks, optimizer_step=False)
        assert (mm["logits"].cpu() - logits.detach()).abs().max() < 1e-3
        product = product2
    assert _check_grads(product.engine_grads(), oracle, tag=f"dropout {mode}") > 150


def build_like(oracle):
    """A copy of the oracle (same weights, no dropout) — a fresh autograd leaf set."""
    import copy
    o = copy.deepcopy(oracle)
    o.enable_dropout(0.0, 0)
    return o


def test_reference_training_loop_on_the_autograd_bridge():
    """The reference's loop body, statement for statement (src/reinforce.py:326-353): rollout -> compute_metrics ->
    (loss / ga).backward() -> clip_grad_value_ -> optim_gpt.step() -> zero_grad(), on the package's objects.  The rollout's
    logprobs / entropies carry a graph whose backward is the engine's; param.grad holds the oracle's gradients in the
    reference layout; the engine-backed torch optimiser leaves the same parameters as `train_iteration` and as
    torch.optim.AdamW on the oracle."""
    from torch.nn.utils import clip_grad
    P, Tn, B = 64, 3, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    product_b, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=43)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(5))
    _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, True, 0.0, 1.0, 0.01)
    oparams = [p for n, p in oracle.named_parameters() if not n.startswith("yolox")]
    before = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    ograds = {n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None}
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    # ---- the reference loop on the bridge ----
    trainer = ja.ReinforceTrainer(cfg, product)
    optim_gpt, optim_yolox = product.configure_optimizers(cfg)
    assert isinstance(optim_gpt, torch.optim.Optimizer) and optim_yolox is None
    product.train()
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    rollout = trainer.rollout(env, forced_actions=forced, start_positions=start)
    assert rollout["logprobs"].grad_fn is not None and rollout["entropies"].grad_fn is not None
    metrics = trainer.compute_metrics(rollout)
    (metrics["loss"] / cfg.gradient_accumulation).backward()
    checked = 0
    for name, p in product.named_parameters():
        if name not in ograds or ograds[name].abs().max() < 1e-12:
            continue
        assert p.grad is not None and p.grad.shape == ograds[name].shape, name
        err = (p.grad.cpu() - ograds[name]).abs().max().item() / ograds[name].abs().max().item()
        assert err < grad_bar(name), (name, err)          # real tensors in the reference's layout
        checked += 1
    assert checked > 150
    clip_grad.clip_grad_value_(product.parameters(), 1)
    assert max(float(p.grad.abs().max()) for p in product.parameters() if p.grad is not None) <= 1.0
    optim_gpt.step()
    optim_gpt.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in product.parameters() if p.grad is not None)
    # ---- the same step through train_iteration and through torch on the oracle ----
    tr_b = ja.ReinforceTrainer(cfg, product_b)
    env_b = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr_b.train_iteration(env_b, forced_actions=forced, start_positions=start, optimizer_step=True)
    product_b.pull_parameters()
    torch.nn.utils.clip_grad_value_(oparams, 1)
    torch.optim.AdamW(oparams, lr=1e-3).step()
    sd_a, sd_b = product.state_dict(), product_b.state_dict()       # bound model: state_dict is current without a pull
    for name, p in oracle.named_parameters():
        g = ograds.get(name)
        if g is None or name.startswith("yolox") or not p.requires_grad:
            continue
        sig = g.abs() > 1e-2 * g.abs().max()              # AdamW's first step ~ lr * sign(g): compare where the sign is defined
        upd_a = (sd_a[name].detach().cpu() - before[name])[sig]
        upd_b = (sd_b[name].detach().cpu() - before[name])[sig]
        upd_o = (p.detach() - before[name])[sig]
        assert torch.allclose(upd_a, upd_o, atol=5e-5), (name, (upd_a - upd_o).abs().max())
        assert torch.allclose(upd_a, upd_b, atol=5e-5), (name, (upd_a - upd_b).abs().max())
    # a second iteration through run(): the engine keeps the AdamW moments, eval-mode numerics follow the new weights
    m = trainer.run(0, 1, 0, batches=[{"image": images.to(DEV), "bboxes": bboxes}], max_iters=2)
    assert torch.isfinite(m["loss"]) and trainer.iter_num == 2
    product.eval()
    with torch.no_grad():
        ro = trainer.rollout(ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True), sample_actions=False)
    assert ro["logprobs"].grad_fn is None and torch.isfinite(ro["logits"]).all()


def test_two_ranks_share_one_gpu_and_average_gradients(tmp_path):
    """two rank processes (gloo, both on cuda:0) each run the REINFORCE iteration on their half of a
    batch; the all-reduced mean gradient equals the mean of the oracle's per-rank gradients (BatchNorm and reward
    statistics stay per rank, as in the reference), and both ranks hold bit-identical parameters after the step."""
    import socket
    import subprocess
    from jolineedle_amd.dist import shard_range
    P, Tn, B, world = 64, 3, 4, 2
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=47)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(9))
    torch.save({"P": P, "T": Tn, "B": B, "images": images, "bboxes": bboxes, "start": start, "forced": forced}, tmp_path / "case.pt")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = str(Path(__file__).resolve().parent / "dist_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path)]) for r in range(world)]
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    out = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    assert all(o["world_seen"] == world for o in out)
    # oracle: per-rank gradients (per-rank BatchNorm statistics), averaged
    _, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    per_rank = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        _oracle_reinforce_grads(oracle, images[lo:hi], bboxes[lo:hi], start[lo:hi], forced[lo:hi], P, Tn, True, 0.25, 1.5, 0.01)
        per_rank.append({n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None})
        oracle.eval()
    checked = 0
    for name, g0 in per_rank[0].items():
        ref = (g0 + per_rank[1][name]) / world
        scale = ref.abs().max().item()
        if scale < 1e-12 or name.startswith("yolox"):
            continue
        tol = grad_bar(name)
        for r in range(world):
            loc = out[r]["local"][name]
            assert (loc - per_rank[r][name]).abs().max().item() < tol * max(per_rank[r][name].abs().max().item(), 1e-12), (name, r)
            assert (out[r]["mean"][name] - ref).abs().max().item() < tol * scale, (name, r)
        checked += 1
    assert checked > 150
    for k, v in out[0]["params"].items():
        # same mean gradient, same AdamW: bit-identical parameters (BatchNorm running statistics stay per rank — the
        # reference's DDP hands out rank 0's, and rank 0 is the one that writes checkpoints: DESIGN.md §5)
        assert torch.equal(v, out[1]["params"][k]), k


def test_split_weight_gradients_stay_close_to_fp32_mfma(tmp_path):
    """The wide 1x1 and dense 3x3 weight gradients run on split-bf16 MFMA products (hi*hi + hi*lo + lo*hi):
    measured directly against the same kernels on fp32 MFMA (JN_WW_EXACT=1 — read once per process, so one process per
    mode; the fused 1x1 kernels' weight-gradient phase uses the same products with a compile-time switch and is held by
    the oracle tests only).  A leaf gradient: the deviation must stay far inside the 1e-3 bar of the oracle tests, and
    everything that is not a weight gradient of those kernels (data path, BatchNorm parameters, the transformer) must
    agree to run-to-run (atomics) noise."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    worker = str(Path(__file__).resolve().parent / "split_grad_worker.py")
    outs = {}
    for mode, env_extra in (("split", {}), ("exact", {"JN_WW_EXACT": "1"})):
        env = dict(os.environ); env.pop("JN_WW_EXACT", None); env.update(env_extra)
        out = tmp_path / f"{mode}.pt"
        assert subprocess.run([sys.executable, worker, str(out)], env=env, timeout=600).returncode == 0
        outs[mode] = torch.load(out)
    worst_w, worst_rest = 0.0, 0.0
    for k, ge in outs["exact"].items():
        scale = ge.abs().max().item()
        if scale < 1e-12:
            continue
        d = (outs["split"][k] - ge).abs().max().item() / scale
        if k.endswith("conv.weight") and not k.endswith(("dconv.conv.weight", "stem.conv.weight")):
            worst_w = max(worst_w, d)
        else:
            worst_rest = max(worst_rest, d)
    assert worst_w < 1e-4, worst_w            # measured 1.5e-5 (two runs of one mode: up to 7e-6, the order of the atomics)
    assert worst_rest < 2e-5, worst_rest      # the data path is untouched: atomics-order noise only


def test_optimizer_step_matches_adamw_with_clip():
    P, Tn, B = 64, 3, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=43)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(5))
    _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, True, 0.0, 1.0, 0.01)
    params = [p for n, p in oracle.named_parameters() if not n.startswith("yolox")]
    before = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    ograds = {n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None}
    opt = torch.optim.AdamW(params, lr=1e-3)
    torch.nn.utils.clip_grad_value_(params, 1)
    opt.step()
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    tr = ja.ReinforceTrainer(cfg, product)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=True)
    product.pull_parameters()
    psd = dict(product.named_parameters())
    for name, p in oracle.named_parameters():
        if name.startswith("yolox") or not p.requires_grad:
            continue
        # AdamW's first step moves a weight by ~lr * sign(g): compare where the sign of g is well defined
        g = ograds.get(name)
        if g is None:
            continue
        sig = g.abs() > 1e-2 * g.abs().max()
        got_upd = (psd[name].detach().cpu() - before[name])[sig]
        ref_upd = (p.detach() - before[name])[sig]
        assert torch.allclose(got_upd, ref_upd, atol=5e-5), (name, (got_upd - ref_upd).abs().max())
        assert ref_upd.abs().max() > 5e-4


# --------------------------------------------------------------------------------------
# detector: yolox-s PAFPN + head + decode + postprocess/NMS
# --------------------------------------------------------------------------------------
def _gap_threshold(scores, k):
    """A threshold in a wide gap of the sorted scores near rank k (robust to 1e-7 score noise)."""
    v = torch.unique(scores.flatten()).flip(0)          # distinct values, descending
    k = min(k, len(v) - 20)
    best, arg = 0.0, k
    for i in range(max(1, k - 15), min(len(v) - 1, k + 15)):
        gap = float(v[i] - v[i + 1]) / float(v[i])
        if gap > best:
            best, arg = gap, i
    assert best > 1e-5
    return float((v[arg] + v[arg + 1]) / 2)


def _same_boxes(got, ref, tol):
    """Same rows up to reordering of near-tied scores: every got row pairs with an unused ref row whose
    score is within 1e-3 and whose (clamped) box is within tol."""
    assert got.shape == ref.shape, (got.shape, ref.shape)
    used = set()
    gs, rs = got[:, 4] * got[:, 5], ref[:, 4] * ref[:, 5]
    for i in range(got.shape[0]):
        cand = [j for j in range(ref.shape[0]) if j not in used and abs(float(gs[i] - rs[j])) < 1e-3]
        assert cand, i
        d = torch.stack([(got[i, :4] - ref[j, :4]).abs().sum() for j in cand])
        k = int(d.argmin())
        assert float(d[k]) < tol, (i, float(d[k]))
        assert (got[i, 4:] - ref[cand[k], 4:]).abs().max() < 1e-3      # obj / cls probabilities (north star 1e-3)
        used.add(cand[k])


def _blocky_images(N, P, seed):
    """Spatially varied content (random-resolution blocks + noise): white-noise images give almost
    position-independent features and thousands of tied scores."""
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(N, 3, P, P)
    for r in (4, 8, 16):
        x += torch.nn.functional.interpolate(torch.rand(N, 3, P // r, P // r, generator=g), size=(P, P), mode="nearest")
    return (x / 3 + 0.1 * torch.rand(N, 3, P, P, generator=g)).clamp(0, 1)


def _detector_pair(P, thr, image_processor="yolox-s", max_batch=4, seed=9, calib=None, **kw):
    product, oracle = make_pair(seed, patch_size=P, block_size=4, image_processor=image_processor,
                                detector_conf_threshold=thr, max_batch=max_batch, max_det_per_patch=512, **kw)
    if calib is not None:                       # BN running statistics of real activations (signal survives the depth)
        bns = [m for m in oracle.yolox.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        oracle.yolox.train()
        with torch.no_grad():
            f = oracle.yolox.backbone(calib)
            h = oracle.yolox.head
            for k in range(3):
                t = h.stems[k](f[k]); h.cls_convs[k](t); h.reg_convs[k](t)
        oracle.yolox.eval()
        for m in bns:
            m.momentum = 0.03
    with torch.no_grad():                       # spread the scores (default-init predictors are almost constant)
        for k in range(3):
            oracle.yolox.head.cls_preds[k].weight.mul_(40.0 if calib is None else 3.0)
            oracle.yolox.head.obj_preds[k].weight.mul_(40.0 if calib is None else 3.0)
            oracle.yolox.head.reg_preds[k].weight.mul_(8.0 if calib is None else 2.0)
    product.load_state_dict(oracle.state_dict())
    return product, oracle


@pytest.mark.parametrize("P,ip", [(64, "yolox-s"), (448, "yolox-s"), (96, "yolox-nano")])
def test_detector_backbone_and_raw_head(P, ip):
    product, oracle = _detector_pair(P, 0.5, ip)
    x = torch.rand((2, 3, P, P), generator=torch.Generator().manual_seed(P))
    with torch.no_grad():
        fpn = oracle.yolox.backbone(x)
        raw = oracle.yolox.head(fpn)                       # [N, A, 6] decoded
    got_fpn = product.backbone_features(x, net=_lib.JN_NET_DETECTOR)
    for i in range(3):
        assert (got_fpn[i].cpu() - fpn[i]).abs().max() < 5e-4, i
    eng = product.engine()
    A = raw.shape[1]
    got_raw = torch.empty((2, A, 6), device=DEV)
    boxes = torch.zeros((2, eng.cfg.max_det_per_patch, 7), device=DEV)
    counts = torch.zeros(2, device=DEV, dtype=torch.int32)
    check(eng.lib.jn_detect(eng.handle, ptr(x.to(DEV)), 2, ptr(boxes), ptr(counts), ptr(got_raw),
                            _lib.current_stream(torch.device(DEV))), "jn_detect")
    torch.cuda.synchronize()
    assert A == sum((P // s) ** 2 for s in (8, 16, 32))
    assert (got_raw.cpu()[..., :4] - raw[..., :4]).abs().max() < 1e-3 * max(1.0, P / 64)   # boxes, pixels
    assert (got_raw.cpu()[..., 4:] - raw[..., 4:]).abs().max() < 1e-5                      # obj / cls probabilities


def test_three_way_split_dense_3x3_is_as_accurate_as_the_fp32_matrix_pipe(monkeypatch):
    """Round 3: the stride-1 dense 3x3 layers of the yolox-s nets run on the bf16 matrix pipe with three-way split operands
    (conv3_x3_kernel) when whole rounds of its 16 x 16 tiles are the cheaper launch.  The eval-mode detector backbone
    over 8 patches of 320 px (40x40 ... 10x10 maps) with the route on and off (JN_NO_CONV3_X3=1) against the oracle in
    fp64: the split route is not further from fp64 than the fp32 pipe (x 1.5 + 1e-6), and it was taken (outputs differ)."""
    P = 320
    product, oracle = _detector_pair(P, 0.5, "yolox-s", max_batch=8)
    x = torch.rand((8, 3, P, P), generator=torch.Generator().manual_seed(23))
    o64 = copy.deepcopy(oracle).double()
    with torch.no_grad():
        want = o64.yolox.backbone(x.double())
    on = [t.cpu().double() for t in product.backbone_features(x, net=_lib.JN_NET_DETECTOR)]
    monkeypatch.setenv("JN_NO_CONV3_X3", "1")
    off = [t.cpu().double() for t in product.backbone_features(x, net=_lib.JN_NET_DETECTOR)]
    monkeypatch.delenv("JN_NO_CONV3_X3")
    assert any(not torch.equal(a, b) for a, b in zip(on, off))
    for a, b, w in zip(on, off, want):
        scale = w.abs().max().item()
        e_on, e_off = (a - w).abs().max().item() / scale, (b - w).abs().max().item() / scale
        assert e_off < 1e-4 and e_on < 1.5 * e_off + 1e-6, (e_on, e_off)


def _bf16_emulated_backbone(net, x):
    """The same graph with every conv's operands (activations, weights) and output rounded to bf16 on the CPU: what
    bf16 storage + bf16 MFMA operands cost on this network, independent of any kernel."""
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    hooks, saved = [], []
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d):
                saved.append((m, m.weight.data.clone()))
                if m.in_channels != 12:                          # the stem reads the fp32 image with fp32 weights
                    m.weight.data.copy_(bf(m.weight.data))
                hooks.append(m.register_forward_pre_hook(lambda mod, inp: (inp[0] if mod.in_channels == 12 else bf(inp[0]),)))
                hooks.append(m.register_forward_hook(lambda mod, inp, out: bf(out)))
        out = net(x)
        for h in hooks:
            h.remove()
        for m, w in saved:
            m.weight.data.copy_(w)
    return out


@pytest.mark.parametrize("P", [96, 448])
def test_bf16_mode_detector_backbone(P):
    """bf16 inference mode of the yolox-s detector (dense 3x3 convs on v_mfma_f32_16x16x32_bf16).  A random-init
    yolox-s is badly conditioned under bf16 (the CPU graph with bf16-rounded conv operands / outputs is itself ~7 % off
    the fp32 one), so the bar is: the HIP path is no further from fp32 than that emulation, and agrees with it."""
    product, oracle = _detector_pair(P, 0.5, "yolox-s", calib=_blocky_images(2, P, 3), act_dtype="bf16")
    x = _blocky_images(2, P, 5)
    with torch.no_grad():
        fpn = oracle.yolox.backbone(x)
    emu = _bf16_emulated_backbone(oracle.yolox.backbone, x)
    got = product.backbone_features(x, net=_lib.JN_NET_DETECTOR)
    for i in range(3):
        g = got[i].cpu()
        e_emu = (emu[i] - fpn[i]).abs().mean().item()
        assert (g - fpn[i]).abs().mean().item() < 1.5 * e_emu + 1e-3 * fpn[i].abs().mean().item(), i
        assert (g - emu[i]).abs().mean().item() < 1.5 * e_emu + 1e-3 * fpn[i].abs().mean().item(), i


@pytest.mark.parametrize("P,keep", [(64, 20), (448, 40)])
def test_detector_postprocess_nms_vs_oracle(P, keep):
    """NeedleYOLOX.forward inference branch: threshold + NMS + clamp; ragged outputs incl. None."""
    from oracle import yolox_ref
    x = _blocky_images(3, P, P + 7)
    _, oracle0 = _detector_pair(P, 0.5, calib=x)
    with torch.no_grad():
        raw = oracle0.yolox.head(oracle0.yolox.backbone(x))
    thr = _gap_threshold(raw[..., 4] * raw[..., 5], 3 * keep)                  # ~keep survivors per patch
    product, oracle = _detector_pair(P, thr, calib=x)
    thr2 = float((raw[2, :, 4] * raw[2, :, 5]).max()) * 1.01
    with torch.no_grad():
        ref_out, ref_fpn, _ = oracle.yolox(x)
    out, fpn_outs, losses = product.yolox(x)
    assert losses == {} and len(out) == 3 and len(fpn_outs) == 3
    for b in range(3):
        if ref_out[b] is None:
            assert out[b] is None
            continue
        got, ref = out[b].cpu(), ref_out[b]
        assert ref.shape[0] <= product.engine().cfg.max_det_per_patch
        _same_boxes(got, ref, 1e-3 * P)       # L1 over 4 coords: 2.5e-4 of the patch size per coordinate
        assert got[:, :4].min() >= 0 and got[:, :4].max() <= P - 1
    # a threshold above every score -> the reference's None
    product2, _ = _detector_pair(P, min(max(thr2, 0.5), 0.9999), calib=x)
    out2, _, _ = product2.yolox(x[2:3])
    assert out2 == [None]


def test_rollout_with_detection_vs_oracle():
    from oracle import env_ref, rollout_ref
    P, Tn, B = 64, 3, 2
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=8)
    images = _blocky_images(B, 3 * P, 8)
    calib = images[:, :, :P, :P].contiguous()
    _, oracle0 = _detector_pair(P, 0.5, max_batch=B, calib=calib)
    with torch.no_grad():
        raw = oracle0.yolox.head(oracle0.yolox.backbone(images[:, :, :P, :P]))
    thr = _gap_threshold(raw[..., 4] * raw[..., 5], 30)
    product, oracle = _detector_pair(P, thr, max_batch=B, calib=calib)
    forced = torch.tensor([[1, 3, 0], [3, 1, 2]])
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), do_detection=True,
                                  forced_actions=forced, start_positions=start)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    ro = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, do_detection=True, forced_actions=forced,
                                                           start_positions=start)
    assert torch.equal(ro["positions"].cpu(), ref["positions"])
    for b in range(B):
        assert len(ro["bboxes"][b]) == len(ref["bboxes"][b]) == Tn + 1
        for t in range(Tn + 1):
            r, g = ref["bboxes"][b][t], ro["bboxes"][b][t]
            assert (r is None) == (g is None), (b, t)
            if r is not None:
                _same_boxes(g.cpu(), r, 1e-3 * P)


@pytest.mark.parametrize("ip,P,N", [("yolox-s", 96, 3), ("yolox-nano", 128, 2)])
def test_detector_training_step_vs_oracle(ip, P, N):
    """NeedleYOLOX.forward(patches, targets) loss branch + backward: SimOTA assignment, the five
    loss terms and every yolox.* gradient (PAFPN, head convs, predictors; train-mode BN) against torch autograd on
    the CPU oracle (the YOLOX package is not in the reference tree: parity is against the restated algorithm)."""
    product, oracle = _detector_pair(P, 0.5, image_processor=ip, max_batch=N)
    g = torch.Generator().manual_seed(31)
    x = _blocky_images(N, P, 8)
    tg = torch.zeros((N, 3, 5))
    tg[0, 0] = torch.tensor([0, 10, 14, 60, 70.])
    tg[1, 0] = torch.tensor([0, 30, 30, 90, 64.])
    tg[1, 1] = torch.tensor([0, 4, 50, 30, 90.])
    if N > 2:
        tg[2, 1] = torch.tensor([0, 40, 8, 80, 40.])          # a padding row BEFORE the box: the published head then
                                                               # takes row 0 (zeros) as the box — reproduced
    import copy
    det64 = copy.deepcopy(oracle.yolox).double().train()        # conditioning reference: the same graph in fp64
    _, _, ref64 = det64(x.double(), tg.double())
    ref64["total_loss"].backward()
    g64 = {n: p.grad for n, p in det64.named_parameters() if p.grad is not None}
    det = oracle.yolox.train()
    det.zero_grad()
    _, _, ref = det(x, tg)
    ref["total_loss"].backward()
    product.engine_zero_grad()
    got = product.yolox.loss_and_backward(x, tg)              # jn_detector_step: loss + backward in one engine call
    for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg"):
        r, v = float(ref[k]), float(got[k])
        assert abs(v - r) < 2e-3 * max(1.0, abs(r)), (k, v, r)
    grads = product.engine_grads("yolox.")
    checked = 0
    for name, p in det.named_parameters():
        if p.grad is None:
            continue
        gp, rg = grads["yolox." + name], p.grad
        scale = rg.abs().max().item()
        if scale < 1e-10:
            assert gp.abs().max().item() < 1e-7, name
            continue
        # the class-branch gradient is sigmoid(logit) - IoU at a handful of anchors: where the two nearly cancel, fp32
        # itself is only good to a few per cent (the fp32 oracle's own distance to fp64 measures that)
        cond = (rg.double() - g64[name]).abs().max().item() / scale
        err = (gp.double() - g64[name]).abs().max().item() / scale
        assert err < max(5e-3, 8.0 * cond), (name, err, cond, scale)
        checked += 1
    assert checked > 100
    # optim_yolox.step(): AdamW with clip_grad_value_(1) on the yolox.* group only
    params = [p for _, p in det.named_parameters()]
    # compare the update where the gradient is well above the parity noise of the (unclipped) gradient
    bigs = {n: p.grad.abs() > 2e-2 * p.grad.abs().max() for n, p in det.named_parameters()}
    torch.nn.utils.clip_grad_value_(params, 1)
    before = {n: p.detach().clone() for n, p in det.named_parameters()}
    torch.optim.AdamW(params, lr=1e-3).step()
    eng = product.engine()
    _lib.check(eng.lib.jn_optimizer_step_group(eng.handle, 1, 1e-3, 0.01, 1.0, 1.0, None), "step")
    product.pull_parameters()
    sd = product.state_dict()
    for name, p in det.named_parameters():
        big = bigs[name]
        if big.any():
            upd_ref, upd = (p.detach() - before[name])[big], (sd["yolox." + name].cpu() - before[name])[big]
            assert (upd - upd_ref).abs().max() < 5e-5, name


def _loose_box_match(got, ref, thr, P, stol=4e-3):
    """Predictions of an eval head that sits on TRAIN-mode FPN maps (batch statistics: maps agree to ~1e-4, bar 1e-3):
    every box whose score clears the threshold by `stol` on one side has a partner on the other (score within stol, box
    within stol * P per coordinate).  Boxes inside the band around the threshold may come or go."""
    def rows(t):
        return [] if t is None else [r for r in t.cpu()]
    for a, b in ((rows(got), rows(ref)), (rows(ref), rows(got))):
        for r in a:
            s = float(r[4] * r[5])
            if s < thr + stol:
                continue
            ok = [q for q in b if abs(float(q[4] * q[5]) - s) < stol and float((q[:4] - r[:4]).abs().max()) < stol * P]
            assert ok, (s, r[:4].tolist())


@pytest.mark.parametrize("shared_encoder", [False, True])
def test_reference_loop_with_detector_loss_on_the_autograd_bridge(shared_encoder):
    """VERDICT round 3, item 1.  The reference's loop body with `detection_enabled` (its default), statement for statement
    (src/reinforce.py:326-353): rollout -> compute_metrics -> get_detection_batch -> `_, _, yolo_loss = yolox(patches_yolox,
    bboxes_yolox)` -> `loss += yolo_loss["total_loss"]` -> `(loss / ga).backward()` with ga = 2 -> clip -> both optimisers.
    `total_loss` carries a graph (yolox.py::_DetectorGraph) next to the rollout's; ONE backward runs both engines' backwards;
    param.grad of yolox.* and of the decision tensors equals the oracle's autograd gradients accumulated over the two
    iterations; `outputs` / `fpn_outs` of the loss-branch call are the eval head on the train-mode maps (src/models/yolox.py
    :74-91).  The detection batch (11 patches) exceeds max_batch = 8: two resident passes.  shared_encoder: no gpt_backbone —
    the detector's own PAFPN encodes the patches (detached), the detector pass then must not disturb the rollout's slots."""
    import copy
    from torch.nn.utils import clip_grad
    from oracle import env_ref, rollout_ref, yolox_ref
    P, Tn, B, ga, MBt = 64, 3, 2, 2, 8
    arch = dict(gpt_backbone=None) if shared_encoder else dict(gpt_backbone="yolox-nano")
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=8)
    images = _blocky_images(B, 3 * P, 8)
    env0 = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    px, bx = env0.get_detection_batch(1, generator=torch.Generator().manual_seed(3))
    px, bx = px.cpu(), bx.cpu().float()
    N = px.shape[0]
    assert N > MBt, N
    chunks = [(i, min(MBt, N - i)) for i in range(0, N, MBt)]

    def build(thr):      # BN running statistics calibrated on the detection batch: the eval head then sees signal, scores spread
        return _detector_pair(P, thr, image_processor="yolox-nano", max_batch=MBt, calib=px, **arch)
    # dry run of iteration 1's detector passes on a throw-away oracle: a threshold inside a wide gap of the eval-head scores
    _, oracle0 = build(0.5)
    dry, scores = oracle0.yolox.train(), []
    for i, n in chunks:
        with torch.no_grad():
            _, f0, _ = dry(px[i:i + n], bx[i:i + n])
            raw = dry.eval().head(f0)
            dry.train()
        scores.append((raw[..., 4] * raw[..., 5]).flatten())
    v = torch.unique(torch.cat(scores)).flip(0)
    gap, arg = max((float(v[i] - v[i + 1]), i) for i in range(8, 120))
    assert gap > 0.012, gap                       # the band of _loose_box_match (2 x 4e-3) fits inside
    thr = float((v[arg] + v[arg + 1]) / 2)
    product, oracle = build(thr)
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=ga)
    cfg.detection_enabled, cfg.yolo_lr = True, 2e-3
    trainer = ja.ReinforceTrainer(cfg, product)
    optim_gpt, optim_yolox = product.configure_optimizers(cfg)
    oracle.train(); oracle.zero_grad()
    det64 = copy.deepcopy(oracle.yolox).double().train()         # conditioning reference of the detector terms (fp64)
    norm = rollout_ref.ReturnNormaliser()
    before = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    for it in range(ga):
        forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(5 + it))
        # ---- the reference's statements on the package's objects (src/reinforce.py:302-341) ----
        trainer.iter_num = it + 1
        product.train()
        env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
        rollout = trainer.rollout(env, forced_actions=forced, start_positions=start)
        metrics = trainer.compute_metrics(rollout)
        loss = metrics["loss"]
        patches_yolox, bboxes_yolox = env.get_detection_batch(1, generator=torch.Generator().manual_seed(3))
        assert torch.equal(patches_yolox.cpu(), px)
        yolox = trainer.yolox_model()
        outputs, fpn_outs, yolo_loss = yolox(patches_yolox, bboxes_yolox)
        total_loss = yolo_loss["total_loss"]
        assert total_loss.grad_fn is not None and rollout["logprobs"].grad_fn is not None
        loss += total_loss
        (loss / cfg.gradient_accumulation).backward()
        # ---- the same on the oracle (per-chunk BatchNorm statistics / normalisation: the documented chunking) ----
        ro = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), forced_actions=forced, start_positions=start)
        loss_o = rollout_ref.reinforce_metrics(ro, 0.01, norm)["loss"]
        ref_out, ref_fpn, yl = [], [[], [], []], 0.0
        for i, n in chunks:
            o, f, lo = oracle.yolox(px[i:i + n], bx[i:i + n])
            ref_out += o
            for lvl in range(3):
                ref_fpn[lvl].append(f[lvl].detach())
            yl = yl + lo["total_loss"] * (n / N)
            _, _, l64 = det64(px[i:i + n].double(), bx[i:i + n].double())
            (l64["total_loss"] * (n / N) / ga).backward()
        assert abs(float(yolo_loss["total_loss"]) - float(yl)) < 2e-3 * max(1.0, abs(float(yl)))
        loss_o = loss_o + yl
        (loss_o / ga).backward()
        assert abs(float(loss) - float(loss_o)) < 2e-3 * max(1.0, abs(float(loss_o)))
        # the rest of NeedleYOLOX.forward: fpn_outs (train-mode maps) and the eval head's predictions on them
        assert len(outputs) == N and len(fpn_outs) == 3
        for lvl in range(3):
            want = torch.cat(ref_fpn[lvl])
            assert fpn_outs[lvl].shape == want.shape and (fpn_outs[lvl].cpu() - want).abs().max() < 1e-3, lvl
        assert sum(o is not None for o in ref_out) >= 2
        for b_ in range(N):
            _loose_box_match(outputs[b_], ref_out[b_], thr, P)
            if outputs[b_] is not None:
                assert outputs[b_][:, :4].min() >= 0 and outputs[b_][:, :4].max() <= P - 1
    # ---- param.grad after the two iterations = the oracle's accumulated autograd gradients ----
    g64 = {"yolox." + n: p.grad for n, p in det64.named_parameters() if p.grad is not None}
    n_det = n_dec = 0
    named = dict(product.named_parameters())
    for name, po in oracle.named_parameters():
        if po.grad is None or po.grad.abs().max() < 1e-12:
            continue
        gp = named[name].grad
        assert gp is not None and gp.shape == po.grad.shape, name
        scale = po.grad.abs().max().item()
        if name.startswith("yolox."):
            if shared_encoder and name.startswith("yolox.backbone."):
                pass                                   # (the policy gradient is detached from it: detector terms only, as g64)
            cond = (po.grad.double() - g64[name]).abs().max().item() / scale
            err = (gp.cpu().double() - g64[name]).abs().max().item() / scale
            assert err < max(5e-3, 8.0 * cond), (name, err, cond)
            n_det += 1
        else:
            err = (gp.cpu() - po.grad).abs().max().item() / scale
            assert err < grad_bar(name), (name, err)
            n_dec += 1
    assert n_det > 100 and n_dec > (20 if shared_encoder else 150), (n_det, n_dec)
    # ---- :343-353: clip, both optimisers; AdamW's first step moves every significant entry by ~ its group's lr ----
    oparams = [p for n, p in oracle.named_parameters() if p.grad is not None]
    clip_grad.clip_grad_value_(product.parameters(), 1)
    optim_gpt.step(); optim_gpt.zero_grad()
    optim_yolox.step(); optim_yolox.zero_grad()
    torch.nn.utils.clip_grad_value_(oparams, 1)
    torch.optim.AdamW([p for n, p in oracle.named_parameters() if p.grad is not None and not n.startswith("yolox")], lr=1e-3).step()
    torch.optim.AdamW([p for n, p in oracle.named_parameters() if p.grad is not None and n.startswith("yolox")], lr=2e-3).step()
    sd = product.state_dict()
    n_upd = 0
    for name, po in oracle.named_parameters():
        if po.grad is None:
            continue
        sig = po.grad.abs() > 5e-2 * po.grad.abs().max()
        if not sig.any():
            continue
        upd, upd_o = (sd[name].detach().cpu() - before[name])[sig], (po.detach() - before[name])[sig]
        assert torch.allclose(upd, upd_o, atol=1e-4), (name, (upd - upd_o).abs().max())
        n_upd += 1
    assert n_upd > 200
    # a second backward through a consumed detector graph, or one whose pass was overwritten, fails loudly
    product.train()
    _, _, l1 = product.yolox(px[:4], bx[:4], predict=False)
    _, _, l2 = product.yolox(px[:4], bx[:4], predict=False)
    with pytest.raises(Exception, match="overwritten|JN_ESTATE|no forward"):
        l1["total_loss"].backward()
    l2["total_loss"].backward()
    with torch.no_grad():
        _, _, l3 = product.yolox(px[:4], bx[:4])
    assert l3["total_loss"].grad_fn is None and torch.isfinite(l3["total_loss"])


def test_reinforce_iteration_with_detector_training():
    """src/reinforce.py:326-353 with detection_enabled: the REINFORCE step and the detector step share one
    iteration; optim_gpt updates everything but yolox.*, optim_yolox (yolo_lr) the detector."""
    P, Tn, B = 64, 3, 2
    product, _ = _detector_pair(P, 0.5, image_processor="yolox-nano", max_batch=8)    # 11 detection patches: two chunks
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=8)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    cfg.detection_enabled, cfg.yolo_lr = True, 2e-3
    before = {k: v.clone() for k, v in product.state_dict().items()}
    tr = ja.ReinforceTrainer(cfg, product)
    m = tr.train_iteration(env, start_positions=start)
    for k in ("loss", "yolo_total_loss", "yolo_iou_loss", "yolo_conf_loss", "yolo_cls_loss", "yolo_l1_loss", "yolo_num_fg"):
        assert k in m and torch.isfinite(torch.as_tensor(float(m[k]))), k
    product.pull_parameters()
    after = product.state_dict()
    kd, kg = "yolox.head.obj_preds.0.weight", "transformer.wte.weight"
    d_det = (after[kd].cpu() - before[kd].cpu()).abs().max().item()
    d_gpt = (after[kg].cpu() - before[kg].cpu()).abs().max().item()
    assert 1e-3 < d_det < 2.5e-3 and 5e-4 < d_gpt < 1.2e-3, (d_det, d_gpt)     # first AdamW step ~ its group's lr


def test_eval_on_batch_detection_metrics_vs_oracle():
    """eval_on_sample of the reference (src/reinforce.py:424-497): greedy rollout with detection, boxes moved to
    full-image coordinates, mAP-50 against the split ground-truth boxes — the GPU rollout's metric equals the one
    computed from the CPU oracle's rollout; the detection batch is a bit-exact gather."""
    from oracle import env_ref, rollout_ref
    from jolineedle_amd import detection
    P, Tn, B = 64, 3, 2
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=8)
    images = _blocky_images(B, 3 * P, 8)
    calib = images[:, :, :P, :P].contiguous()
    _, oracle0 = _detector_pair(P, 0.5, max_batch=B, calib=calib)
    with torch.no_grad():
        raw = oracle0.yolox.head(oracle0.yolox.backbone(images[:, :, :P, :P]))
    thr = _gap_threshold(raw[..., 4] * raw[..., 5], 30)
    product, oracle = _detector_pair(P, thr, max_batch=8, calib=calib)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    env.reset(start)
    tr = ja.ReinforceTrainer(_cfg(T=Tn), product)
    ro0 = tr.rollout(env, do_detection=True, sample_actions=False, start_positions=start)
    m = tr.eval_on_batch(ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True), do_detection=True, merge_bboxes=True)
    for k in ("map", "yolo_map", "loss", "prop_patches_found", "prop_bbox_found"):
        assert k in m, k
    assert 0.0 <= float(m["map"]) <= 1.0 and 0.0 <= float(m["yolo_map"]) <= 1.0
    # the same bookkeeping applied to the oracle's rollout (same greedy trajectory as ro0) gives the same mAP
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), do_detection=True,
                                  sample_actions=False, start_positions=ro0["positions"][:, 0].cpu())
    if torch.equal(ref["positions"], ro0["positions"].cpu()):
        offs = ref["positions"][:, :, [1, 0]] * P
        preds = detection.merge_boxes_batched(detection.patch_bboxes2full_image(ref["bboxes"], offs, ref["masks"]))
        tg = detection.merge_boxes_batched(detection.detection_targets(bboxes, 3, 3, P), target=True)
        ref_map = float(detection.compute_detection_metrics(preds, tg)["map"])
        assert abs(ref_map - float(ja.ReinforceTrainer(_cfg(T=Tn), product).eval_on_batch(
            ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True), True, True)["map"])) < 0.05
    pats, tgt = env.get_detection_batch(sample_neg=1, generator=torch.Generator().manual_seed(1))
    assert pats.shape[1:] == (3, P, P) and tgt.shape[0] == pats.shape[0] and tgt.shape[2] == 5
    loc, msk = env.parse_bboxes()
    ys, xs = torch.nonzero(msk[0].any(-1).cpu())[0].tolist()
    assert torch.equal(pats[0].cpu(), images[0, :, ys * P:(ys + 1) * P, xs * P:(xs + 1) * P])


# --------------------------------------------------------------------------------------
# supervised teacher-forced step (BASELINE configs 1-2)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,P,stop_w", [(2, 4, 64, 0.1), (4, 8, 64, 1.0), (4, 8, 448, 1.0)])     # last: BASELINE configs[0]/[1] sizes
def test_supervised_step_vs_oracle(B, T, P, stop_w):
    product, oracle = make_pair(13, patch_size=P, block_size=T, with_detector=False, image_processor=None,
                                max_batch=B * T)
    patches, cur, positions = synth_tokens(B, T, P, 9, 5, seed=21)
    g = torch.Generator().manual_seed(4)
    nxt = torch.randint(0, 9, (B, T), generator=g)
    nxt[0, 1] = 8                                             # a STOP target exercises the class weight
    masks = torch.ones((B, T), dtype=torch.long)
    masks[1, T - 2:] = 0                                      # padded tail
    w = torch.ones(9); w[8] = stop_w
    keep = masks.flatten() == 1
    # class ids as the supervised loop passes them (src/supervised.py:852, 866): a repeated id (its embed_class row collects
    # two sequences' gradients), the table's last row, and 0
    classes = torch.tensor([3, 99, 3, 0][:B]) if P < 448 else torch.zeros(B, dtype=torch.long)

    def run_oracle(o, dt):
        o.train()
        o.zero_grad()
        lg, _ = o(patches.to(dt), cur, classes, positions)
        ce_ = torch.nn.functional.cross_entropy(lg.reshape(B * T, 9), nxt.flatten(), weight=w.to(dt), reduction="none")
        ls = ce_[keep].mean()
        ls.backward()
        return lg, ls
    ref64 = probe = probe_l2 = logits64 = None
    if P >= 448:                                         # see _check_grads
        import copy
        o64 = copy.deepcopy(oracle).double()
        logits64, _ = run_oracle(o64, torch.float64)
        ref64 = _grads64(o64)
    logits, loss = run_oracle(oracle, torch.float32)
    acc = (logits.reshape(B * T, 9).argmax(1)[keep] == nxt.flatten()[keep]).float().mean()
    cfg = ja.CfgNode(stop_enabled=True, stop_weight=stop_w, learning_rate=1e-3, gradient_accumulation=1)
    tr = ja.SupervisedTrainer(cfg, product)
    m = tr.train_step(patches, cur, nxt, positions, masks, optimizer_step=False, classes=classes)
    assert (m["logits"].cpu() - logits.detach()).abs().max() < 1e-3        # train-mode BN over B*T patches
    if P < 448:                                          # the gradient of a class token lands on ITS row of the table
        ge = product.engine_grads()["embed_class.weight"]
        rows = ge.abs().amax(dim=1) > 0
        assert rows.nonzero().flatten().tolist() == sorted(set(classes.tolist()))
    assert abs(float(m["loss"]) - float(loss)) < 2e-4
    assert abs(float(m["action_accuracy"]) - float(acc)) < 1e-6
    assert abs(float(m["episode_length"]) - float(masks.sum(1).float().mean())) < 1e-6
    if logits64 is not None:            # the forward is pinned far below the 1e-3 bar where the gradient bars lean on the probe
        assert (m["logits"].cpu().double() - logits64.detach()).abs().max() < 1e-4
    grads = product.engine_grads()
    if ref64 is None:
        n = _check_grads(grads, oracle, skip_prefix=(), tag=f"supervised B={B} T={T} P={P}")
    else:
        # the probe's noise level is measured (the engine's train-mode maps of these B * T patches against fp64), its draws
        # come on demand (_check_grads_probed)
        eps = _measured_probe_eps(product, oracle, patches.flatten(0, 1))
        n, draws = _check_grads_probed(grads, oracle, lambda o: run_oracle(o, torch.float32), ref64, eps, skip_prefix=(),
                                       tag=f"supervised B={B} T={T} P={P}", max_draws=12)
    assert n > 150


# --------------------------------------------------------------------------------------
# bf16 inference mode (bf16 activation storage + bf16 MFMA, eval-mode BN): north-star bar 1e-3 on logits
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,N", [(64, 3), (448, 2)])
def test_bf16_mode_backbone_maps(P, N):
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None,
                                act_dtype="bf16")
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(P))
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
    got = product.backbone_features(x)
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        assert err < 5e-3 * ref[i].abs().max().item() + 1e-4, (i, err)      # bf16: ~0.4 % per rounding


def test_bf16_mode_logits_and_rollout_golden(golden):
    g, g4 = golden("g3_gpt_forward.npz"), golden("g4_rollout.npz")
    product, _ = make_pair(int(g["seed"]), int(g["bn_seed"]), patch_size=64, block_size=6,
                           image_processor="yolox-nano", gpt_backbone="yolox-nano", act_dtype="bf16")
    patches, actions, positions = synth_tokens(3, 6, 64, 9, 5, seed=int(g["tok_seed"]))
    lg, emb = product(patches, actions, torch.zeros(3, dtype=torch.long), positions)
    assert np.abs(lg.cpu().numpy() - g["full_logits"]).max() < 1e-3
    assert np.abs(emb.cpu().numpy() - g["full_emb"]).max() < 1e-3
    P, Tn = int(g4["P"]), int(g4["T"])
    images, bboxes, _ = synth_batch(4, 4, 5, P, seed=int(g4["batch_seed"]))
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    ro = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, sample_actions=False, start_positions=T_(g4["start"]))
    for k in ("masks", "logit_masks", "positions"):
        assert np.array_equal(ro[k].cpu().numpy(), g4[k]), k
    assert np.array_equal(ro["rewards"].cpu().numpy(), g4["rewards"])
    for k in ("returns", "logprobs", "entropies"):
        assert np.allclose(ro[k].cpu().numpy(), g4[k], atol=1e-3), k
    pos = ro["positions"].cpu()
    for b in range(4):                                     # the gather stays bit exact in every mode
        y, x = pos[b, 1].tolist()
        assert torch.equal(ro["patches"][b, 1].cpu(), images[b, :, y * P:(y + 1) * P, x * P:(x + 1) * P])


def test_rollout_with_detection_shared_encoder_is_race_free():
    """gpt_backbone=None (the reference's default, main.py --gpt-backbone): the detector's own PAFPN encodes the patches
    (src/models/gpt.py:376-380), so the per-glimpse detector pass and the next encoder pass use the same workspace and
    must stay in stream order.  Three identical rollouts must agree with each other bit for bit and with the oracle."""
    from oracle import env_ref, rollout_ref
    P, Tn, B = 64, 3, 2
    images, bboxes, start = synth_batch(B, 3, 3, P, seed=8)
    images = _blocky_images(B, 3 * P, 8)
    calib = images[:, :, :P, :P].contiguous()
    _, oracle0 = _detector_pair(P, 0.5, max_batch=B, calib=calib, gpt_backbone=None)
    with torch.no_grad():
        raw = oracle0.yolox.head(oracle0.yolox.backbone(images[:, :, :P, :P]))
    thr = _gap_threshold(raw[..., 4] * raw[..., 5], 30)
    product, oracle = _detector_pair(P, thr, max_batch=B, calib=calib, gpt_backbone=None)
    forced = torch.tensor([[1, 3, 0], [3, 1, 2]])
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), do_detection=True,
                                  forced_actions=forced, start_positions=start)
    runs = []
    for _ in range(3):
        env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
        runs.append(ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, do_detection=True, forced_actions=forced,
                                                                      start_positions=start))
    ro = runs[0]
    assert torch.equal(ro["positions"].cpu(), ref["positions"])
    assert (ro["logits"].cpu() - ref["logits"]).abs().max() < 1e-3
    for other in runs[1:]:
        assert torch.equal(other["logits"], ro["logits"])
        assert torch.equal(other["det_counts"], ro["det_counts"])
    for b in range(B):
        for t in range(Tn + 1):
            r, g = ref["bboxes"][b][t], ro["bboxes"][b][t]
            assert (r is None) == (g is None), (b, t)
            if r is not None:
                _same_boxes(g.cpu(), r, 1e-3 * P)
                for other in runs[1:]:
                    assert torch.equal(other["bboxes"][b][t], g)


def test_reinforce_gradients_stop_at_a_detached_detector_encoder():
    """gpt_backbone=None: the reference detaches the detector's FPN maps ("Do not backpropagate through yolox",
    src/models/gpt.py:376-380), so the policy gradient reaches embed_fpn and the transformer only and every yolox.*
    gradient stays zero."""
    P, Tn, B = 64, 3, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, image_processor="yolox-nano", gpt_backbone=None)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=41)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3))
    ro, m = _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, True, 0.25, 1.5, 0.01)
    tr = ja.ReinforceTrainer(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1), product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    got_m = tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    assert abs(float(got_m["loss"]) - float(m["loss"])) < 2e-4
    grads = product.engine_grads()
    n = _check_grads(grads, oracle, tag="reinforce detached")
    assert n > 30
    for name, g in grads.items():
        if name.startswith("yolox"):
            assert float(g.abs().max()) == 0.0, name
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for n_, p in oracle.named_parameters() if n_.startswith("yolox"))


def test_bf16_mode_supervised_forward_at_config2_size():
    """BASELINE configs[1] ("same config on 1 x MI355X, bf16"): the teacher-forced forward of the supervised step at its
    sizes (B=4, T=8, 448 px) in the bf16 inference mode — logits within 1e-3 of the fp32 CPU oracle (north star).  bf16
    TRAINING is a refused deviation (test_bf16_mode_refuses_training)."""
    B, T, P = 4, 8, 448
    product, oracle = make_pair(13, patch_size=P, block_size=T, with_detector=False, image_processor=None,
                                max_batch=B, act_dtype="bf16")
    patches, cur, positions = synth_tokens(B, T, P, 9, 10, seed=23)
    with torch.no_grad():
        ref, ref_emb = oracle(patches, cur, torch.zeros(B, dtype=torch.long), positions)
    lg, emb = product(patches, cur, torch.zeros(B, dtype=torch.long), positions)
    assert lg.shape == (B, T, 9)
    assert (lg.cpu() - ref).abs().max() < 1e-3
    assert (emb.cpu() - ref_emb).abs().max() < 1e-3


def test_bf16_mode_refuses_training():
    product, _ = make_pair(5, patch_size=64, block_size=3, with_detector=False, image_processor=None, act_dtype="bf16")
    images, bboxes, start = synth_batch(2, 3, 3, 64, seed=1)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, 64, 3, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=3, learning_rate=1e-3, gradient_accumulation=1), product)
    with pytest.raises(_lib.JnError, match="fp32"):
        tr.train_iteration(env, start_positions=start)


# --------------------------------------------------------------------------------------
# teacher trajectories: indexed gather bit exact; collated sample == reference's
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("nimg,C,G,P,N", [(3, 3, (4, 5), 16, 11), (2, 3, (2, 2), 448, 5), (4, 1, (3, 3), 7, 9), (1, 3, (2, 3), 64, 0)])
def test_indexed_gather_bit_exact(nimg, C, G, P, N):
    from jolineedle_amd.trajectory import gather_indexed
    g = torch.Generator().manual_seed(nimg * 10 + P)
    images = torch.rand((nimg, C, G[0] * P, G[1] * P), generator=g).to(DEV)
    ii = torch.randint(-1, nimg, (N,), generator=g)
    pos = torch.stack((torch.randint(0, G[0], (N,), generator=g), torch.randint(0, G[1], (N,), generator=g)), 1)
    out = gather_indexed(images, ii, pos, P)
    assert out.shape == (N, C, P, P)
    for n in range(N):
        y, x = pos[n].tolist()
        want = images[ii[n], :, y * P:(y + 1) * P, x * P:(x + 1) * P] if ii[n] >= 0 else torch.zeros((C, P, P), device=DEV)
        assert torch.equal(out[n], want), n
    if N:
        with pytest.raises(AssertionError):
            gather_indexed(images, ii.clamp(min=0), pos + G[0] + G[1], P)


def test_generate_sample_matches_reference_walks(golden):
    import random
    from jolineedle_amd.trajectory import NeedleSimpleEnv, assemble_samples
    g8 = golden("g8_trajectories.npz")
    for name in (str(n) for n in g8["names"]):
        P, gh, gw, seed, pyseed, T, kmin, kmax, binom = (int(v) for v in g8[f"{name}.args"])
        n = 3 * gh * P * gw * P
        image = (torch.arange(n, dtype=torch.float32) / n).reshape(3, gh * P, gw * P).to(DEV)
        start = g8[f"{name}.start"]
        env = NeedleSimpleEnv(image, P, g8[f"{name}.boxes"], seed=seed)
        random.seed(pyseed)
        s = env.generate_sample(T, kmin, kmax, bool(binom), None if start[0] < 0 else (int(start[0]), int(start[1])))
        for k in ("patches", "patches_yolox", "positions", "current_actions", "next_actions", "labels", "masks",
                  "local_bboxes", "bboxes_yolox"):
            want = g8[f"{name}.{k}"]
            assert tuple(s[k].shape) == want.shape, (name, k, s[k].shape, want.shape)
            assert np.array_equal(s[k].cpu().numpy(), want), (name, k)


def test_supervised_iteration_on_generated_trajectories():
    """End to end on the device: collate layout -> teacher walks -> teacher-forced step + detector step -> AdamW on
    both groups.  The step itself is checked against the oracle above; here the feeding and the update are."""
    P, T, B = 64, 6, 3
    product, _ = make_pair(9, patch_size=P, block_size=T, image_processor="yolox-nano", gpt_backbone="yolox-nano",
                           max_batch=B * T)
    images, bboxes, _ = synth_batch(B, 4, 4, P, seed=3)
    batch = {"image": images.to(DEV), "bboxes": bboxes, "class_id": torch.zeros(B, dtype=torch.long)}
    cfg = ja.CfgNode(patch_size=P, max_seq_len=T, min_keypoints=0, max_keypoints=1, binomial_keypoints=False,
                     stop_enabled=True, stop_weight=1.0, learning_rate=1e-3, yolo_lr=1e-3, gradient_accumulation=1,
                     detection_enabled=True)
    tr = ja.SupervisedTrainer(cfg, product)
    t = tr.generate_trajectories(batch, seed=5)
    assert t["patches"].shape == (B, T, 3, P, P) and t["patches_yolox"].shape[0] == t["bboxes_yolox"].shape[0]
    assert t["bboxes_yolox"].shape[1:] == (3, 6) and t["local_bboxes"].shape == (B, T, 3, 6)
    pos, msk = t["positions"].cpu(), t["masks"].cpu()
    for b in range(B):
        for s in range(T):
            y, x = pos[b, s].tolist()
            want = images[b, :, y * P:(y + 1) * P, x * P:(x + 1) * P] * msk[b, s]
            assert torch.equal(t["patches"][b, s].cpu(), want)
    before = {k: v.clone() for k, v in product.state_dict().items() if k.endswith("weight")}
    m = tr.train_iteration(batch, seed=5)
    assert np.isfinite(float(m["loss"])) and np.isfinite(float(m["yolo_total_loss"])) and float(m["yolo_num_fg"]) >= 0
    product.pull_parameters()
    after = product.state_dict()
    moved_gpt = sum(int(not torch.equal(before[k], after[k])) for k in before if not k.startswith("yolox"))
    moved_det = sum(int(not torch.equal(before[k], after[k])) for k in before if k.startswith("yolox"))
    assert moved_gpt > 50 and moved_det > 50
    m2 = tr.train_iteration(batch, seed=5)                      # same walks again: the loss went down
    assert float(m2["action_loss"]) < float(m["action_loss"])


def test_reference_supervised_loop_on_the_autograd_bridge():
    """The reference's supervised loop body in structure (src/supervised.py:863-868, 897-902): ``action_logits, _ =
    model(patches, current_actions, classes=classes, positions=positions)`` in train mode -> its cross-entropy
    (compute_metrics, :138-177) -> ``loss.backward()`` -> ``optim_gpt.step()``; ``optim_gpt.zero_grad()`` — NO clipping.
    The logits carry a graph whose backward is the engine's; param.grad equals the oracle's autograd gradients in the
    reference layout; the parameters after the step equal ``train_step``'s (in-engine loss) and torch.optim.AdamW's on the
    oracle; a pass over the encoder between forward and backward makes the backward fail loudly instead of computing on
    overwritten activations."""
    B, T, P, stop_w = 2, 4, 64, 0.3
    mk = lambda: make_pair(13, patch_size=P, block_size=T, with_detector=False, image_processor=None, max_batch=B * T)
    product, oracle = mk()
    product_b, _ = mk()
    patches, cur, positions = synth_tokens(B, T, P, 9, 5, seed=21)
    nxt = torch.randint(0, 9, (B, T), generator=torch.Generator().manual_seed(4))
    nxt[0, 1] = 8
    masks = torch.ones((B, T), dtype=torch.long)
    masks[1, T - 1:] = 0
    w = torch.ones(9); w[8] = stop_w
    keep = masks.flatten() == 1
    oracle.train(); oracle.zero_grad()
    classes = torch.tensor([5, 61])                       # batch["class_id"] of a multi-class dataset (src/dataset.py:289-295)
    lg, _ = oracle(patches, cur, classes, positions)
    loss_o = torch.nn.functional.cross_entropy(lg.reshape(B * T, 9), nxt.flatten(), weight=w, reduction="none")[keep].mean()
    loss_o.backward()
    ograds = {n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None}
    before = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    cfg = ja.CfgNode(stop_enabled=True, stop_weight=stop_w, learning_rate=1e-3, gradient_accumulation=1)
    # ---- the reference loop on the bridge ----
    trainer = ja.SupervisedTrainer(cfg, product)
    optim_gpt, optim_yolox = product.configure_optimizers(cfg)
    assert optim_yolox is None
    product.train()
    dev = lambda t: t.to(DEV)
    action_logits, embeddings = product(dev(patches), dev(cur), classes=dev(classes), positions=dev(positions))
    assert action_logits.grad_fn is not None and action_logits.shape == (B, T, 9) and embeddings.shape == (B, T + 1, product.n_embd)
    assert (action_logits.detach().cpu() - lg.detach()).abs().max() < 1e-3
    metrics = trainer.compute_metrics(action_logits, dev(nxt), dev(masks))
    assert abs(float(metrics["loss"]) - float(loss_o)) < 2e-4
    loss = metrics["loss"]
    loss.backward()
    checked = 0
    for name, p in product.named_parameters():
        if name not in ograds or ograds[name].abs().max() < 1e-12:
            continue
        assert p.grad is not None and p.grad.shape == ograds[name].shape, name
        err = (p.grad.cpu() - ograds[name]).abs().max().item() / ograds[name].abs().max().item()
        assert err < grad_bar(name), (name, err)
        checked += 1
    assert checked > 150
    optim_gpt.step()
    optim_gpt.zero_grad()
    # ---- the same step inside the engine (train_step) and through torch on the oracle: no clipping anywhere ----
    ja.SupervisedTrainer(cfg, product_b).train_step(patches, cur, nxt, positions, masks, optimizer_step=True, classes=classes)
    product_b.pull_parameters()
    oparams = [p for n, p in oracle.named_parameters()]
    torch.optim.AdamW(oparams, lr=1e-3).step()
    sd_a, sd_b = product.state_dict(), product_b.state_dict()
    n_cmp = 0
    for name, p in oracle.named_parameters():
        g = ograds.get(name)
        if g is None or not p.requires_grad:
            continue
        sig = g.abs() > 1e-2 * g.abs().max()
        upd_a = (sd_a[name].detach().cpu() - before[name])[sig]
        upd_b = (sd_b[name].detach().cpu() - before[name])[sig]
        upd_o = (p.detach() - before[name])[sig]
        assert torch.allclose(upd_a, upd_o, atol=5e-5), (name, (upd_a - upd_o).abs().max())
        assert torch.allclose(upd_a, upd_b, atol=5e-5), (name, (upd_a - upd_b).abs().max())
        n_cmp += 1
    assert n_cmp > 150
    # ---- stale activations fail loudly ----
    logits2, _ = product(dev(patches), dev(cur), classes=torch.zeros(B, dtype=torch.long), positions=dev(positions))
    with torch.no_grad():
        product.backbone_features(patches[:, 0])                 # an eval pass over the encoder's workspace in between
    with pytest.raises(Exception, match="overwr|stale|JN_ESTATE|no supervised forward"):
        logits2.sum().backward()
    # eval mode / no_grad keep the graph-free eval numerics
    product.eval()
    lg_eval, _ = product(dev(patches), dev(cur), torch.zeros(B, dtype=torch.long), dev(positions))
    assert lg_eval.grad_fn is None
    # run(): two iterations of the loop on generated trajectories (single rank)
    prod_c, _ = make_pair(5, patch_size=P, block_size=T, image_processor="yolox-nano", gpt_backbone="yolox-nano", max_batch=4 * T)
    images, bboxes, _ = synth_batch(4, 3, 4, P, seed=31)
    rcfg = ja.CfgNode(patch_size=P, max_seq_len=T, min_keypoints=0, max_keypoints=2, binomial_keypoints=False, stop_enabled=True,
                      stop_weight=1.0, learning_rate=1e-3, yolo_lr=1e-3, gradient_accumulation=1, detection_enabled=True, max_iters=2)
    tr_c = ja.SupervisedTrainer(rcfg, prod_c)
    sd0 = {k: v.detach().cpu().clone() for k, v in prod_c.state_dict().items() if k.endswith("weight")}
    m = tr_c.run(0, 1, 0, batches=[{"image": images.to(DEV), "bboxes": bboxes}], seed=5)
    assert torch.isfinite(m["loss"]) and torch.isfinite(m["yolo_total_loss"]) and tr_c.iter_num == 2
    sd1 = prod_c.state_dict()
    assert sum(int(not torch.equal(sd0[k], sd1[k].cpu())) for k in sd0 if k.startswith("yolox")) > 50
    assert sum(int(not torch.equal(sd0[k], sd1[k].cpu())) for k in sd0 if not k.startswith("yolox")) > 50


def test_rollout_graph_survives_a_dropped_rollout_dict_and_refuses_stale_state():
    """ADVICE (round 2): ``loss = compute_metrics(rollout(env))["loss"]`` drops the rollout dict before backward — the graph
    node keeps the buffers the engine reads through raw pointers alive; an eval rollout between forward and backward
    restarts the state the backward reads, which then fails with JN_ESTATE instead of silently wrong gradients."""
    import gc
    P, Tn, B = 64, 3, 2
    product, oracle = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=43)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(5))
    _oracle_reinforce_grads(oracle, images, bboxes, start, forced, P, Tn, True, 0.0, 1.0, 0.01)
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    trainer = ja.ReinforceTrainer(cfg, product)
    product.train()
    mk_env = lambda: ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    loss = trainer.compute_metrics(trainer.rollout(mk_env(), forced_actions=forced.clone(), start_positions=start.clone()))["loss"]
    gc.collect()
    junk = [torch.full((B, Tn + 1, product.n_embd), 7.0, device=DEV) for _ in range(64)]     # would land in freed blocks
    junk += [torch.full((B, Tn, 9), 7.0, device=DEV) for _ in range(64)]
    loss.backward()
    del junk
    n = 0
    for name, p in oracle.named_parameters():
        if p.grad is None or name.startswith("yolox") or p.grad.abs().max() < 1e-12:
            continue
        got = dict(product.named_parameters())[name].grad.cpu()
        assert (got - p.grad).abs().max().item() < grad_bar(name) * p.grad.abs().max().item(), name
        n += 1
    assert n > 150
    ro = trainer.rollout(mk_env(), forced_actions=forced, start_positions=start)
    product.eval()
    with torch.no_grad():
        trainer.rollout(mk_env(), sample_actions=False)          # eval rollout in between
    with pytest.raises(Exception, match="JN_ESTATE|preceding|-4|overwritten"):
        trainer.compute_metrics(ro)["loss"].backward()


def test_two_ranks_supervised_loop_averages_gradients(tmp_path):
    """src/supervised.py:812-911 is the reference's only DistributedDataParallel site: two rank processes (gloo, both on
    cuda:0) run the supervised loop body on their half of a batch — ``model(...)`` in train mode, cross-entropy,
    ``loss.backward()``, ``optim.step()`` (ONE all-reduce of the flat gradient buffer inside, no clipping).  Each rank's
    local gradient equals the oracle's on its trajectories (per-rank BatchNorm statistics), the mean gradient equals the
    mean of the oracle's, both ranks hold bit-identical parameters after the step, and a further ``run()`` iteration works."""
    import socket
    import subprocess
    P, T, B, world = 64, 4, 4, 2
    images, bboxes, _ = synth_batch(B, 3, 4, P, seed=57)
    torch.save({"P": P, "T": T, "B": B, "images": images, "bboxes": bboxes, "mode": "supervised"}, tmp_path / "case.pt")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = str(Path(__file__).resolve().parent / "dist_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path)]) for r in range(world)]
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    out = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    assert all(o["world_seen"] == world for o in out)
    _, oracle = make_pair(5, patch_size=P, block_size=T, with_detector=False, image_processor=None)
    per_rank = []
    for r in range(world):
        tr = out[r]["trajectories"]
        oracle.train(); oracle.zero_grad()
        lg, _ = oracle(tr["patches"], tr["current_actions"], torch.zeros(tr["patches"].shape[0], dtype=torch.long), tr["positions"])
        keep = tr["masks"].flatten() == 1
        ce_ = torch.nn.functional.cross_entropy(lg.reshape(-1, 9), tr["next_actions"].flatten(), reduction="none")
        ce_[keep].mean().backward()
        per_rank.append({n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None})
        oracle.eval()
    checked = 0
    for name, g0 in per_rank[0].items():
        ref = (g0 + per_rank[1][name]) / world
        scale = ref.abs().max().item()
        if scale < 1e-12 or name.startswith("yolox"):
            continue
        tol = grad_bar(name)
        for r in range(world):
            assert (out[r]["local"][name] - per_rank[r][name]).abs().max().item() < tol * max(per_rank[r][name].abs().max().item(), 1e-12), (name, r)
            assert (out[r]["mean"][name] - ref).abs().max().item() < tol * scale, (name, r)
        checked += 1
    assert checked > 150
    for k, v in out[0]["params"].items():
        assert torch.equal(v, out[1]["params"][k]), k
    assert all(o["run_iters"] == 2 and o["run_loss_finite"] for o in out)


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_multi_rank_launch_reports_every_rank(ranks):
    """`bench.py --gpus N` as a fresh child process (the parent has not touched the GPU; it starts one rank process per
    GPU as main.py:428-433 spawns its ranks): rehearsed on ONE GPU over gloo (JN_BENCH_BACKEND / JN_BENCH_SAME_DEVICE).
    Rank 0's JSON line carries every rank: n_ranks_seen, the global batch, a finite whole-job value over the slowest rank's
    time, and the diagnostics a first real multi-GPU run will be read by: each rank's own time per
    step, its all-reduce time (events around the ONE collective of an iteration), forward / backward section times.
    (N = 8 is not rehearsed here: the GPU box allows 6 processes on its card, this one included.)"""
    import json
    import os
    import subprocess
    env = dict(os.environ, JN_BENCH_BACKEND="gloo", JN_BENCH_SAME_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    root = Path(__file__).resolve().parent.parent
    per_rank = 16 // ranks
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch",
                        str(per_rank), "--grid", "3", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["n_ranks_seen"] == ranks and out["config"]["global_batch"] == 16
    assert out["scaling"] == "weak" and np.isfinite(out["value"]) and out["value"] > 0
    # whole-job value = patches of ALL ranks over the slowest rank's time
    assert abs(out["value"] - 16 * 20 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]
    mr = out["multi_rank"]
    assert len(mr["rank_ms_per_step"]["per_rank"]) == ranks and len(mr["allreduce_ms"]["per_rank"]) == ranks
    assert 0 < mr["rank_ms_per_step"]["min"] <= mr["rank_ms_per_step"]["max"] <= out["ms_per_step"] * 1.001
    assert all(0 < v < out["ms_per_step"] for v in mr["allreduce_ms"]["per_rank"])
    assert 5_000_000 < mr["allreduce_ms"]["bytes"] < 6_500_000      # the 1.37 M floats of SURVEY §8e (padded arena)
    assert all(v > 0 for v in mr["forward_ms_per_pass_per_rank"] + mr["backward_ms_per_step_per_rank"])


# --------------------------------------------------------------------------------------
# detection augmentation: the fused pass == the op chain of the oracle for the same parameters
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,P", [(5, 64), (3, 448), (4, 37), (2, 4)])
def test_augment_fused_pass_vs_oracle(N, P):
    from jolineedle_amd.augment import DetectionAugment
    from oracle.augment_ref import augment_ref
    g = torch.Generator().manual_seed(N * 7 + P)
    x = torch.rand((N, 3, P, P), generator=g)
    noise = torch.randn((N, 3, P, P), generator=g)
    aug = DetectionAugment(planckian_coeffs=torch.tensor([[1.25, 0.7], [0.85, 1.3], [1.05, 0.95]]), p_planckian=0.7, p_gray=0.4,
                           p_blur=0.7, p_noise=0.6, p_motion=0.6, seed=N + P)
    prm = aug.sample_params(N)
    prm[0, 3], prm[0, 4], prm[0, 5] = 0.5, 0.25, 0.05                  # make sure every op is exercised at least once
    prm[0, 6:15] = torch.tensor([0.2, 0.1, 0.0, 0.1, 0.2, 0.1, 0.0, 0.1, 0.2])
    prm[-1, 0], prm[-1, 1], prm[-1, 2] = 1.4, 0.6, 1.0
    from jolineedle_amd.augment import plasma_stretch
    prm[0, 15], prm[0, 16], prm[0, 17] = -0.15, 0.45, 0.6              # plasma shadow on the first patch
    prm[0, 18] = plasma_stretch(prm[0:1, 17])[0]
    got = aug(x.to(DEV), params=prm, noise=noise.to(DEV)).cpu()
    want = augment_ref(x, prm, noise, seed=aug.last_seed)
    bad = (got - want).abs() > 2e-6
    # the shadow mask is a threshold on a fractal: a pixel whose value is within an fp32 rounding of the quantity may fall
    # on the other side on the device (fma contraction) — at most a handful per patch
    assert int(bad.sum()) <= max(3, int(1e-4 * got.numel())), int(bad.sum())
    if P >= 37:
        sh = (got[0] - augment_ref(x[:1], torch.cat((prm[:1, :15], torch.zeros(1, 5)), 1), noise[:1])[0]).abs() > 1e-4
        assert 0.02 < float(sh.float().mean()) < 0.9                   # the shadow really darkened part of the patch
    ident = torch.zeros_like(prm); ident[:, 0] = ident[:, 1] = ident[:, 3] = ident[:, 10] = 1.0
    assert torch.equal(aug(x.to(DEV), params=ident).cpu(), x)          # undrawn ops are exact identities


def test_augment_device_noise_statistics_and_trainer_hook():
    from jolineedle_amd.augment import DetectionAugment
    aug = DetectionAugment(p_planckian=0, p_gray=0, p_blur=0, p_noise=1.0, p_motion=0, p_shadow=0, noise_std=0.05, seed=9)
    x = torch.full((4, 3, 256, 256), 0.5, device=DEV)
    d = (aug(x) - x).cpu()
    assert abs(float(d.mean())) < 2e-4 and abs(float(d.std()) - 0.05) < 5e-4
    k = float(((d / 0.05) ** 4).mean())                                # kurtosis of a normal = 3
    assert abs(k - 3.0) < 0.1
    assert abs(float((d[:, :, :, :-1] * d[:, :, :, 1:]).mean())) / 0.05 ** 2 < 0.02      # neighbours uncorrelated
    d2 = (aug(x) - x).cpu()
    assert not torch.equal(d, d2)                                      # a new field per call
    # trainer hook: opt-in, applied to the detector patches of a REINFORCE iteration
    product, _ = make_pair(3, patch_size=64, block_size=3, image_processor="yolox-nano", gpt_backbone="yolox-nano")
    images, bboxes, start = synth_batch(2, 3, 3, 64, seed=1)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, 64, 3, 1, True)
    tr = ja.ReinforceTrainer(_cfg(T=3, learning_rate=1e-3, gradient_accumulation=1, detection_enabled=True), product)
    assert getattr(tr, "detection_augment", None) is None
    tr.init_detection()
    m = tr.train_iteration(env, start_positions=start)
    assert tr.detection_augment.calls == 1 and np.isfinite(float(m["yolo_total_loss"]))


# --------------------------------------------------------------------------------------
# checkpoints and infer
# --------------------------------------------------------------------------------------
def test_checkpoint_round_trip_and_detection_checkpoint(tmp_path):
    product, oracle = make_pair(4, patch_size=64, block_size=4, image_processor="yolox-nano", gpt_backbone="yolox-nano")
    patches, actions, positions = synth_tokens(2, 4, 64, 9, 5, seed=3)
    lg0, _ = product(patches, actions, torch.zeros(2, dtype=torch.long), positions)
    # a reference-style checkpoint (DDP prefixes, optimiser entries) written by hand from the oracle's weights
    ck = {"model": {"module." + k: v for k, v in oracle.state_dict().items()}, "optimizer-gpt": {"state": {}}, "optimizer-yolox": {}}
    torch.save(ck, tmp_path / "checkpoint_best.pt")
    other, _ = make_pair(5, patch_size=64, block_size=4, image_processor="yolox-nano", gpt_backbone="yolox-nano")
    lg_other, _ = other(patches, actions, torch.zeros(2, dtype=torch.long), positions)
    assert (lg_other - lg0).abs().max() > 1e-3
    ja.load_checkpoint(ja.CfgNode(resume_training=str(tmp_path)), other, best=True)
    lg1, _ = other(patches, actions, torch.zeros(2, dtype=torch.long), positions)
    assert torch.equal(lg1, lg0)
    # our own checkpoint: engine-side (trained) weights are pulled before saving
    path = ja.save_checkpoint(other, tmp_path / "mine")
    saved = torch.load(path, map_location="cpu", weights_only=False)
    assert set(saved) >= {"model", "optimizer-gpt", "optimizer-yolox"} and set(saved["model"]) == set(oracle.state_dict())
    # detection checkpoint: only yolox.* is replaced
    third, o3 = make_pair(6, patch_size=64, block_size=4, image_processor="yolox-nano", gpt_backbone="yolox-nano")
    ja.load_detection_checkpoint(ja.CfgNode(detection_checkpoint=str(tmp_path / "checkpoint_best.pt")), third)
    sd3, sd0 = third.state_dict(), oracle.state_dict()
    assert all(torch.equal(sd3[k], sd0[k]) for k in sd0 if k.startswith("yolox."))
    assert any(not torch.equal(sd3[k], sd0[k]) for k in sd0 if k.startswith("transformer."))
    x = torch.rand(2, 3, 64, 64)
    assert torch.equal(third.yolox(x)[1][2], product.yolox(x)[1][2])      # same detector features now


def test_checkpoint_resume_continues_the_optimiser_state(tmp_path):
    """main.py:436-449 / 532-563: "optimizer-gpt" carries torch.optim.AdamW state dicts.  Two iterations in one go equal
    one iteration -> checkpoint -> fresh model + load -> one iteration (moments and bias-correction step restored), the
    saved dict loads into a real torch.optim.AdamW over the same parameter list, and a frozen detector backbone
    (--freeze-image-processor, src/models/gpt.py:264-268) is left alone by optim_yolox."""
    from jolineedle_amd import checkpoint
    P, Tn, B = 64, 3, 2
    images, bboxes, start = synth_batch(B, 3, 4, P, seed=51)
    forced = torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(6))
    cfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    cfg.resume_training = str(tmp_path)

    def iteration(model):
        tr = ja.ReinforceTrainer(cfg, model)
        env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
        tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=True)
    a, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    iteration(a); iteration(a)
    a.pull_parameters()
    b, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    iteration(b)
    checkpoint.save_checkpoint(b, tmp_path)
    ck = torch.load(tmp_path / "checkpoint.pt", weights_only=False)
    st = ck["optimizer-gpt"]["state"]
    assert len(st) > 150 and all(float(v["step"]) == 1.0 for v in st.values())
    c, _ = make_pair(6, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)     # other weights: everything comes from the file
    checkpoint.load_checkpoint(cfg, c)
    iteration(c)
    c.pull_parameters()
    # without the moments the second step differs (AdamW's bias correction restarts at step 1, exp_avg at zero)
    d, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    d.load_state_dict(ck["model"])
    iteration(d)
    d.pull_parameters()
    sa, sc, sd_, s1 = a.state_dict(), c.state_dict(), d.state_dict(), ck["model"]
    # compare the SECOND step's update per tensor in the L2 sense (an element whose gradient is rounding noise may take a
    # different sign under AdamW's normalisation; the atomics make that noise differ from run to run)
    n_ok, errs = 0, []
    for k, p in a.named_parameters():
        if not p.requires_grad or k.startswith("yolox"):
            continue
        upd_a = sa[k].cpu() - s1[k]
        if float(upd_a.norm()) < 1e-9:
            continue
        err_c = float((sc[k].cpu() - s1[k] - upd_a).norm() / upd_a.norm())
        err_d = float((sd_[k].cpu() - s1[k] - upd_a).norm() / upd_a.norm())
        errs.append((err_c, k))
        n_ok += err_d > 3 * max(err_c, 0.02)
    errs.sort()
    assert errs[len(errs) // 2][0] < 0.05, errs[len(errs) // 2]          # the typical tensor repeats to a few per cent (0.02 - 0.032 over ten runs)
    assert errs[int(0.95 * len(errs))][0] < 0.1 and errs[-1][0] < 0.5, errs[-5:]   # the noisiest (tiny gradients) stay bounded
    assert n_ok > 100
    # the dict is a torch.optim.AdamW state dict for the same parameter list
    plist = [torch.nn.Parameter(p.detach().cpu().clone()) for n, p in b.named_parameters() if not n.startswith("yolox")]
    topt = torch.optim.AdamW(plist, lr=1e-3)
    topt.load_state_dict(ck["optimizer-gpt"])
    assert len(topt.state_dict()["state"]) == len(st)
    # a REFERENCE checkpoint's optimiser entry (torch.optim.AdamW over the reference's parameter list, real moments) lands on
    # the right tensors: positions agree because named_parameters() walks the reference's module order
    e_model, e_oracle = make_pair(8, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    oparams = [p for n, p in e_oracle.named_parameters() if not n.startswith("yolox")]
    gen = torch.Generator().manual_seed(12)
    for p_ in oparams:
        p_.grad = torch.randn(p_.shape, generator=gen) * 0.01
    ref_opt = torch.optim.AdamW(oparams, lr=3e-4)
    ref_opt.step()
    ref_sd = ref_opt.state_dict()
    mine, _ = e_model.configure_optimizers(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1))
    with pytest.warns(UserWarning, match="param_names"):          # positions are all an untagged file has: announced
        mine.load_state_dict(ref_sd)
    back = mine.state_dict()
    assert mine.param_groups[0]["lr"] == 3e-4
    onames = [n for n, p in e_oracle.named_parameters() if not n.startswith("yolox")]
    n_eq = 0
    for i, st in ref_sd["state"].items():
        if not oparams[i].requires_grad or i not in back["state"]:
            continue
        assert torch.allclose(back["state"][i]["exp_avg"], st["exp_avg"], atol=1e-9), onames[i]
        assert torch.allclose(back["state"][i]["exp_avg_sq"], st["exp_avg_sq"], atol=1e-12), onames[i]
        n_eq += 1
    assert n_eq > 150
    # a file in ANOTHER list order (what rounds 1-2 of this package wrote: engine execution order, same shapes inside a CSP
    # pair) lands on the right tensors BY NAME: positions reversed, names kept
    pnames = back["param_groups"][0]["param_names"]
    assert pnames == onames and len(set(pnames)) == len(pnames)
    n_p = len(pnames)
    rev = {"state": {n_p - 1 - i: st for i, st in back["state"].items()},
           "param_groups": [dict(back["param_groups"][0], params=list(range(n_p)), param_names=pnames[::-1])]}
    g_model, _ = make_pair(9, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    other, _ = g_model.configure_optimizers(_cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                               # no shape-mismatch skips, no "untagged" notice
        other.load_state_dict(rev)
    back2 = other.state_dict()
    assert set(back2["state"]) == set(back["state"])
    for i, st in back["state"].items():
        assert torch.equal(back2["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(back2["state"][i]["exp_avg_sq"], st["exp_avg_sq"]), pnames[i]
    with pytest.raises(ValueError, match="does not have"):
        other.load_state_dict({"state": {0: back["state"][next(iter(back["state"]))]},
                               "param_groups": [dict(back["param_groups"][0], params=[0], param_names=["no.such.tensor"])]})
    # a checkpoint written without optimisers carries the learning rate the engine last stepped with, never 0
    assert ck["optimizer-gpt"]["param_groups"][0]["lr"] == 1e-3
    # frozen detector backbone: optim_yolox moves the head only
    f, _ = make_pair(7, patch_size=P, block_size=Tn, image_processor="yolox-nano", gpt_backbone="yolox-nano", freeze_image_processor=True)
    before = {kk: v.detach().cpu().clone() for kk, v in f.state_dict().items()}
    fcfg = _cfg(T=Tn, learning_rate=1e-3, gradient_accumulation=1)
    fcfg.detection_enabled, fcfg.yolo_lr = True, 1e-3
    ja.ReinforceTrainer(fcfg, f).train_iteration(ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True), start_positions=start)
    f.pull_parameters()
    after = f.state_dict()
    assert torch.equal(after["yolox.backbone.backbone.dark3.0.pconv.conv.weight"].cpu(), before["yolox.backbone.backbone.dark3.0.pconv.conv.weight"])
    assert not torch.equal(after["yolox.head.stems.0.conv.weight"].cpu(), before["yolox.head.stems.0.conv.weight"])
    assert not any(p.requires_grad for n, p in f.named_parameters() if n.startswith("yolox.backbone."))


def test_infer_images_pads_and_maps_boxes_to_the_full_image():
    P, T = 64, 5
    product, _ = make_pair(8, patch_size=P, block_size=T, image_processor="yolox-nano", gpt_backbone="yolox-nano",
                           detector_conf_threshold=0.2)
    cfg = ja.CfgNode(patch_size=P, max_seq_len=T, stop_enabled=True, detection_enabled=True, entropy_weight=0.01,
                     reward_norm=False, seed=2)
    tr = ja.ReinforceTrainer(cfg, product)
    g = torch.Generator().manual_seed(5)
    images = [(torch.rand((3, 150, 200), generator=g) * 255).to(torch.uint8), torch.rand((3, 128, 64), generator=g)]
    targets = [[[20, 30, 90, 100]], None]
    out = ja.infer_images(tr, images, targets, sample_actions=False)
    assert len(out["boxes"]) == 2 and out["steps"][0] <= T and len(out["duration_ms"]) == 2
    assert {"returns", "episode_length", "prop_patches_found", "map"} <= set(out["metrics"])
    for img, boxes, pos in zip(images, out["boxes"], out["positions"]):
        H, W = -(-img.shape[1] // P) * P, -(-img.shape[2] // P) * P
        assert (pos[:, 0] < H // P).all() and (pos[:, 1] < W // P).all()
        if boxes is not None:
            assert boxes.shape[1] == 7 and (boxes[:, 0] >= 0).all() and (boxes[:, 2] <= W).all() and (boxes[:, 3] <= H).all()
    # plumbing: the boxes of the first visited patch are the detector's boxes on that patch of the PADDED [0, 1] image,
    # shifted by the patch origin (the detector itself is checked against the oracle elsewhere)
    pad = ja.pad_to_patch_multiple((images[0].float() / 255)[None], P)
    y0, x0 = out["positions"][0][0].tolist()
    det = product.yolox(pad[:, :, y0 * P:(y0 + 1) * P, x0 * P:(x0 + 1) * P])[0][0]
    assert det is not None and out["boxes"][0] is not None
    want = det.cpu().clone()
    want[:, [0, 2]] += x0 * P
    want[:, [1, 3]] += y0 * P
    got = out["boxes"][0][: len(want)].cpu()                 # same set; boxes of equal score (zero padding) may swap places
    assert ((got[:, None, :] - want[None, :, :]).abs().max(dim=2).values.min(dim=0).values < 2e-3).all()


# --------------------------------------------------------------------------------------
# edge shapes of the rollout: single agent, single step, 1 x 1 grid, immediate STOP, longest sequence
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,G,Tn,actions", [
    (1, (1, 1), 1, [[3]]),                      # one agent on a one-patch image, one step (every move is clipped)
    (1, (2, 3), 4, [[8, 0, 0, 0]]),             # STOP first: sticky, the episode ends after one step
    (2, (1, 4), 6, [[1, 1, 1, 1, 1, 1], [0, 0, 1, 8, 1, 1]]),   # a one-row image: walks into the border, late STOP
    (3, (3, 3), 20, None),                      # a long sequence (block_size 20) with random moves
])
def test_rollout_edge_shapes_vs_oracle(B, G, Tn, actions):
    from oracle import env_ref, rollout_ref
    P = 64
    product, oracle = make_pair(11, patch_size=P, block_size=Tn, with_detector=False, image_processor=None, max_batch=B)
    images, bboxes, start = synth_batch(B, G[0], G[1], P, seed=B * 10 + Tn)
    forced = (torch.tensor(actions) if actions is not None
              else torch.randint(0, 8, (B, Tn), generator=torch.Generator().manual_seed(3)))
    with torch.no_grad():
        ref = rollout_ref.rollout(oracle, env_ref.EnvRef(images, bboxes, P, Tn, 1, True), forced_actions=forced,
                                  start_positions=start, stop_early=True)
    env = ja.NeedleGeneralEnv(images.to(DEV), bboxes, P, Tn, 1, True)
    ro = ja.ReinforceTrainer(_cfg(T=Tn), product).rollout(env, forced_actions=forced, start_positions=start)
    S = ref["rewards"].shape[1]
    assert ro["rewards"].shape[1] == S and 1 <= S <= Tn
    for k in ("masks", "logit_masks", "positions", "actions"):
        assert torch.equal(ro[k].cpu(), ref[k]), k
    assert torch.equal(ro["rewards"].cpu(), ref["rewards"])
    for k in ("returns", "logprobs", "entropies", "logits"):
        assert (ro[k].cpu() - ref[k]).abs().max() < TOL_LOGIT, k
    assert torch.equal(ro["patches"].cpu(), ref["patches"])


# --------------------------------------------------------------------------------------
# the persistent narrow 1x1 kernel only takes maps of >= 65536 pixels by itself (B = 64 at 448 px): force it onto the
# small parity shapes (ragged last tile, fewer tiles than workgroups) and at its natural size on a 6-patch batch
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,N,train", [(64, 3, False), (96, 5, True), (160, 3, True), (448, 6, False)])
def test_narrow_pointwise_kernel_forced_on_small_maps(P, N, train, monkeypatch):
    if P != 448:
        monkeypatch.setenv("JN_PWN_MIN_M", "1")
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None, max_batch=N)
    x = torch.rand((N, 3, P, P), generator=torch.Generator().manual_seed(P + N))
    oracle.gpt_backbone.train(train)
    with torch.no_grad():
        ref = oracle.gpt_backbone(x)
    got = product.backbone_features(x, train=train)
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        assert err < (1e-3 if train else TOL_MAP), (i, err)
    if train:                                                   # gradients through the same maps (saved statistics)
        oracle.zero_grad()
        outs = oracle.gpt_backbone(x)
        gs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(7 + i)) for i, o in enumerate(outs)]
        sum((o * g).sum() for o, g in zip(outs, gs)).backward()
        product.engine_zero_grad()
        product.backbone_features(x, train=True)
        product.backbone_backward(x, gs)
        grads = product.engine_grads("gpt_backbone.")
        for name, p in oracle.gpt_backbone.named_parameters():
            if p.grad is None:
                continue
            scale = p.grad.abs().max().item() + 1e-6
            assert (grads["gpt_backbone." + name] - p.grad).abs().max().item() / scale < 5e-3, name


def test_wide_data_gradient_large_map_variant_forced_on_small_shapes(monkeypatch):
    """The data gradient of the wide 1x1 layers switches to 128-pixel workgroups above 65536 pixels (B = 64 x 20 steps):
    force that variant onto a parity-sized backward."""
    monkeypatch.setenv("JN_PW_WT_SMALL_M", "0")
    P, N = 96, 3
    product, oracle = make_pair(3, patch_size=P, block_size=6, with_detector=False, image_processor=None)
    g = torch.Generator().manual_seed(23)
    x = torch.rand((N, 3, P, P), generator=g)
    net = oracle.gpt_backbone.train()
    outs = net(x)
    R = [torch.randn(o.shape, generator=g) for o in outs]
    net.zero_grad()
    sum((o * r).sum() for o, r in zip(outs, R)).backward()
    product.engine_zero_grad()
    product.backbone_features(x, train=True)
    product.backbone_backward(x, R)
    got = product.engine_grads("gpt_backbone.")
    for name, p in net.named_parameters():
        scale = p.grad.abs().max().item() + 1e-6
        assert (got["gpt_backbone." + name] - p.grad).abs().max().item() / scale < 2e-3, name
