"""Oracle restatement of the REINFORCE rollout, returns and loss.

TEST INFRASTRUCTURE ONLY.  Follows src/reinforce.py:73-90 (sample_from_logits),
:108-215 (rollout), :92-106 and :217-265 (reward normalisation window and
compute_metrics).  Pinned by tests/golden/make_golden.py (G4, G5 vectors).

Deviation switch: ``forced_actions`` replays a given action sequence
(teacher forcing) so sampled trajectories can be compared step by step; the
reference only has sample / argmax (reinforce.py:83-86).
"""
from typing import Dict, List, Optional

import torch


def sample_from_logits(logits: torch.Tensor, take_best_action: bool = False,
                       forced: Optional[torch.Tensor] = None):
    last = logits[:, -1, :]
    logp = last - last.logsumexp(dim=-1, keepdim=True)        # Categorical(logits=...) normalisation
    probs = logp.exp()
    if forced is not None:
        actions = forced
    elif take_best_action:
        actions = last.argmax(dim=1)
    else:
        actions = torch.multinomial(probs, 1).squeeze(1)
    logprobs = logp.gather(1, actions[:, None]).squeeze(1)
    # torch.distributions.Categorical.entropy clamps logits at finfo.min before p*logp
    entropies = -(logp.clamp(min=torch.finfo(logp.dtype).min) * probs).sum(-1)
    return actions, logprobs, entropies


def discounted_returns(rewards: torch.Tensor, masks: torch.Tensor):
    """reinforce.py:191-202.  masks [B, S+1] -> logit_masks [B, S]; returns [B, S]."""
    logit_masks = torch.roll(masks[:, 1:], shifts=1, dims=(1,))
    logit_masks[:, 0] = True
    returns = torch.flip(torch.cumsum(torch.flip(rewards * logit_masks, dims=(1,)), dim=1), dims=(1,))
    return logit_masks, returns


def rollout(model, env, do_detection: bool = False, sample_actions: bool = True,
            forced_actions: Optional[torch.Tensor] = None, start_positions=None,
            stop_early: bool = True) -> Dict[str, torch.Tensor]:
    B = env.batch_size
    actions = torch.zeros((B, 1), dtype=torch.long)
    classes = torch.zeros((B,), dtype=torch.long)
    patches, infos = env.reset(start_positions)
    positions = infos["positions"].unsqueeze(1)
    bboxes: List[list] = [[] for _ in range(B)]
    masks = [torch.ones(B, dtype=torch.bool)]
    rewards, logprobs, entropies, logits_all = [], [], [], []
    if do_detection:
        # reinforce.py:144 indexes patches[0] (only valid for B == 1); the intent —
        # detect on the start patch of every image — is what is restated.
        out, _, _ = model.yolox(patches[:, 0], None)
        for i in range(B):
            bboxes[i].append(out[i])
    emb = None
    for t in range(env.max_ep_len):
        logits, emb = model(patches, actions, classes, positions, emb)
        forced = None if forced_actions is None else forced_actions[:, t]
        new_actions, lp, ent = sample_from_logits(logits, not sample_actions, forced)
        new_patches, r, terminated, truncated, infos = env.step(new_actions)
        if do_detection:
            out, _, _ = model.yolox(new_patches[:, 0], None)
            for i in range(B):
                bboxes[i].append(out[i])
        rewards.append(r); logprobs.append(lp); entropies.append(ent); masks.append(~terminated)
        logits_all.append(logits[:, -1])
        actions = torch.cat((actions, new_actions[:, None]), dim=1)
        patches = torch.cat((patches, new_patches), dim=1)
        positions = torch.cat((positions, infos["positions"][:, None]), dim=1)
        if stop_early and bool(torch.all(terminated | truncated)):
            break
    rewards = torch.stack(rewards, 1)
    masks = torch.stack(masks, 1)
    logit_masks, returns = discounted_returns(rewards, masks)
    return {"rewards": rewards, "returns": returns, "logprobs": torch.stack(logprobs, 1),
            "entropies": torch.stack(entropies, 1), "masks": masks, "logit_masks": logit_masks,
            "positions": positions, "bboxes": bboxes, "patches": patches,
            "actions": actions[:, 1:], "logits": torch.stack(logits_all, 1)}


class ReturnNormaliser:
    """reinforce.py:68-71, 92-106: mean/std of the masked returns of the PREVIOUS
    optimiser window (init 0, 1; torch.std is the unbiased estimator)."""

    def __init__(self):
        self.values: List[torch.Tensor] = []
        self.mean, self.std = 0, 1

    def roll(self):
        v = torch.cat(self.values) if self.values else torch.zeros(0)
        if len(v) == 0:
            self.mean, self.std = 0, 1
        elif len(v) == 1:
            self.mean, self.std = v[0], 1
        else:
            self.mean, self.std = v.mean(), v.std()
        self.values = []


def reinforce_metrics(ro: Dict[str, torch.Tensor], entropy_weight: float = 0.01,
                      norm: Optional[ReturnNormaliser] = None) -> Dict[str, torch.Tensor]:
    returns, masks = ro["returns"], ro["logit_masks"]
    if norm is not None:
        norm.values.append(returns[masks].clone().detach())
        adv = (returns - norm.mean) / (norm.std + 1e-8)
    else:
        adv = returns
    m = {}
    m["action_loss"] = -(ro["logprobs"] * adv * masks).sum() / masks.sum()
    m["entropy_loss"] = -(ro["entropies"] * masks).sum() / masks.sum()
    m["loss"] = m["action_loss"] + entropy_weight * m["entropy_loss"]
    m["returns"] = (ro["rewards"] * masks).sum(dim=1).mean()
    m["episode_length"] = masks.sum(dim=1).float().mean()
    return m
