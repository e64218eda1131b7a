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
