"""Seeded synthetic inputs shared by tests, bench.py and tests/golden/make_golden.py."""
import torch


def randomize_bn(model, seed):
    """Non-trivial BN statistics / affine so that BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


def synth_batch(B, G_h, G_w, P, seed, nb_max=3):
    """images [B,3,G_h*P,G_w*P] f32 in [0,1); bboxes [B,nb_max,4] int64 xyxy zero-row
    padded (padded_collate_fn layout, src/dataset.py:337-341); start positions [B,2] (y,x)."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand((B, 3, G_h * P, G_w * P), generator=g)
    bboxes = torch.zeros((B, nb_max, 4), dtype=torch.long)
    for b in range(B):
        nb = int(torch.randint(1, nb_max + 1, (1,), generator=g))
        for k in range(nb):
            w = int(torch.randint(P // 8, P, (1,), generator=g))
            h = int(torch.randint(P // 8, P, (1,), generator=g))
            x = int(torch.randint(0, G_w * P - w, (1,), generator=g))
            y = int(torch.randint(0, G_h * P - h, (1,), generator=g))
            bboxes[b, k] = torch.tensor([x, y, x + w, y + h])
    pos = torch.stack((torch.randint(0, G_h, (B,), generator=g),
                       torch.randint(0, G_w, (B,), generator=g)), 1)
    return images, bboxes, pos


def synth_tokens(B, T, P, nA, grid, seed):
    g = torch.Generator().manual_seed(seed)
    patches = torch.rand((B, T, 3, P, P), generator=g)
    actions = torch.randint(0, nA, (B, T), generator=g)
    positions = torch.randint(0, grid, (B, T, 2), generator=g)
    return patches, actions, positions


from jolineedle_amd.config import model_config  # noqa: E402,F401  (kept importable from here for the tests)


def make_pair(seed=0, bn_seed=5, max_batch=8, **kw):
    """(product GPT on the GPU engine, oracle GPTRef on CPU) holding the same weights."""
    import jolineedle_amd as ja
    from oracle.gpt_ref import build_gpt_ref
    okw = dict(kw)
    okw.pop("max_det_per_patch", None)
    okw.pop("act_dtype", None)
    oracle = build_gpt_ref(seed, **okw)
    if bn_seed is not None:
        randomize_bn(oracle, bn_seed)
    oracle.eval()
    product = ja.GPT(model_config(**kw), max_batch=max_batch)
    product.load_state_dict(oracle.state_dict())
    return product, oracle
