"""Criteria with the reference's call signatures (``utils/losses.py:10-43``) for the drop-in path, where the
model returns log-probs to torch.  The native trainer (training.py) uses the fused HIP criterion instead."""
import torch
from torch import nn


class LanguageModelCriterion(nn.Module):
    def forward(self, input, target, mask):
        target = target[:, : input.size(1)]
        mask = mask[:, : input.size(1)]
        output = -input.gather(2, target.unsqueeze(2)).squeeze(2) * mask
        return torch.sum(output) / torch.sum(mask)


class RewardCriterion(nn.Module):
    def forward(self, input, mask, reward):
        input = input.contiguous().view(-1)
        reward = reward.contiguous().view(-1).unsqueeze(1)
        mask = mask.float()
        output = -input * (mask * reward).contiguous().view(-1)
        return torch.sum(output) / torch.sum(mask)


class LabelSmoothing(nn.Module):
    """``LabelSmoothing`` (utils/losses.py:46-77): masked mean over tokens of KL(q || exp(input)), q = 1 - smoothing on the
    target and smoothing / (V - 1) elsewhere.  Written in closed form instead of materialising q and calling KLDivLoss:
    KL = c (log c - x_t) + s (V - 1) log s - s (sum_v x_v - x_t), with 0 log 0 = 0."""

    def __init__(self, size=0, padding_idx=0, smoothing=0.0):
        super().__init__()
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing

    def forward(self, input, target, mask):
        target = target[:, : input.size(1)]
        mask = mask[:, : input.size(1)].reshape(-1).to(input.dtype)
        x = input.reshape(-1, input.size(-1))
        V = x.size(1)
        xt = x.gather(1, target.reshape(-1, 1)).squeeze(1)
        c, s = self.confidence, self.smoothing / (V - 1)
        xlogx = lambda p: p * torch.log(torch.tensor(p, dtype=x.dtype)).item() if p > 0 else 0.0
        kl = xlogx(c) - c * xt + (V - 1) * xlogx(s) - s * (x.sum(1) - xt)
        return (kl * mask).sum() / mask.sum()
