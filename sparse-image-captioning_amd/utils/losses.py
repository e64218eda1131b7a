"""Criteria with the reference's call signatures (``utils/losses.py:10-77``) for the drop-in path, where the model returns
log-probs to torch.  The native trainer (training.py) uses the fused HIP criterion instead."""
import math

import torch
from torch import nn


def _picked(logp, target):
    """log-prob of every target token: (R, T, V), (R, >=T) -> (R, T)."""
    T = logp.size(1)
    return torch.gather(logp, 2, target[:, :T, None]).squeeze(2)


class LanguageModelCriterion(nn.Module):
    """Masked mean negative log-likelihood: ``-(sum_rt logp[r,t,target] * mask) / sum(mask)`` (losses.py:32-43)."""

    def forward(self, input, target, mask):
        w = mask[:, : input.size(1)]
        return -(_picked(input, target) * w).sum() / w.sum()


class RewardCriterion(nn.Module):
    """SCST loss on the sampled tokens' log-probs: ``-(sum logp * mask * reward) / sum(mask)``, one reward per sampled caption
    (losses.py:10-29; ``input`` (N, ns, L) or (N*ns, L), ``mask`` the same shape, ``reward`` (N*ns,))."""

    def forward(self, input, mask, reward):
        m = mask.float().reshape(reward.numel(), -1)
        lp = input.reshape(reward.numel(), -1)
        return -(lp * m * reward.reshape(-1, 1)).sum() / m.sum()


class LabelSmoothing(nn.Module):
    """``LabelSmoothing`` (utils/losses.py:46-77): masked mean over tokens of KL(q || exp(input)), q = 1 - smoothing on the
    target and smoothing / (V - 1) elsewhere.  Written in closed form instead of materialising q and calling KLDivLoss:
    KL = c (log c - x_t) + s (V - 1) log s - s (sum_v x_v - x_t), with 0 log 0 = 0."""

    def __init__(self, size=0, padding_idx=0, smoothing=0.0):
        super().__init__()
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing

    def forward(self, input, target, mask):
        w = mask[:, : input.size(1)].reshape(-1).to(input.dtype)
        x = input.reshape(-1, input.size(-1))
        V = x.size(1)
        xt = _picked(input, target).reshape(-1)
        c, s = self.confidence, self.smoothing / (V - 1)
        xlogx = lambda p: p * math.log(p) if p > 0 else 0.0
        kl = xlogx(c) - c * xt + (V - 1) * xlogx(s) - s * (x.sum(1) - xt)
        return (kl * w).sum() / w.sum()
