"""ORT with masked weights — mirror of ``sparse_caption/models/relation_transformer_prune.py:116-172`` and
``pruning/masked_layer.py:20-174``: every >=2-D weight ``W`` has a sibling ``W_pruning_mask`` parameter and the
kernels see ``s * W`` (s = bernoulli(sigmoid(m)) in training, round(sigmoid(m)) in eval for supermasks; the
binary mask itself for magnitude / SNIP pruning).  Dropout is 0.1/3 (relation_transformer_prune.py:41,65,96,121).

The mask logits live in a second flat arena with the SAME offsets as the weights, so ``s * W`` for all 147 masked
tensors is one HIP launch (``ortk_mask_apply``) and the straight-through backward another (``ortk_mask_bwd``).
Positions that are not maskable (biases, LayerNorm) hold a neutral logit (sample == 1, zero gradient).

Decoding deviation (documented in DESIGN.md): the reference's `_prune` class never enables cached attention
(its attention modules lack the ``incremental_decoding`` attribute), and ``scripts/eval_model.py:64-88`` decodes
pruned checkpoints through the DENSE class on densified weights.  ``mode="sample"`` here does exactly that:
cached-attention decoding on ``round(sigmoid(m)) * W``.
"""
import ctypes as C

import torch

from . import register_model
from .. import _lib as L
from ..pruning import prune
from ..pruning.prune import PruningMixin
from .relation_transformer import RelationTransformerModel as _Dense

NEUTRAL_LOGIT = 1.0e4   # sigmoid -> exactly 1.0f: round == bernoulli == 1, derivative == 0


@register_model("relation_transformer_prune")
class RelationTransformerModel(PruningMixin, _Dense):
    DROPOUT = 0.1 / 3
    MASKED = True

    def __init__(self, config, precision=None):
        self._mask_flat = None
        self._weff = None
        self._init_pruning_args = (config.prune_type, config.prune_mask_freeze_scope)
        _Dense.__init__(self, config, precision)
        self._init_pruning(*self._init_pruning_args)
        mt = self.mask_type
        self._supermask = mt in prune.SUPER_MASKS
        self.mask_init_value = float(config.prune_supermask_init) if self._supermask else 1.0
        self._mask_flat = torch.full((self._n_train,), NEUTRAL_LOGIT if self._supermask else 1.0)
        self._bind()
        self.reset_masks()
        trainable = self._supermask or mt == prune.SNIP     # masked_layer.py:52-67
        for _, m in self.all_pruning_masks():
            m.requires_grad = trainable
        self._sparsity_coef = None
        self._n_neutral = self._n_train - self.total_mask_params

    # ---- arena plumbing
    def _extra_param_specs(self, entry):
        return [("_pruning_mask", "_mask_flat")] if entry["kind"] == 1 else []

    def _arenas(self):
        return {"": "_flat", "mask": "_mask_flat", "weff": "_weff"}

    @torch.no_grad()
    def reset_masks(self):
        for _, m in self.all_pruning_masks():
            m.fill_(self.mask_init_value)

    def _mode(self, train):
        if not self._supermask:
            return 2
        return 1 if train else 0

    def _eff_params_ptr(self, train, seed):
        lib = L.lib()
        if self._weff is None or self._weff.device != self._flat.device:
            self._weff = torch.empty_like(self._flat)
        if self._n_all > self._n_train:
            self._weff[self._n_train:].copy_(self._flat[self._n_train:])   # the `pe` buffer
        draws = self._draws(train)
        if draws is not None:
            L.check(lib.ortk_mask_apply_draws(L.ptr(self._flat), L.ptr(self._mask_flat), L.ptr(draws), L.ptr(self._weff), self._n_train,
                                              L.stream_ptr()), "ortk_mask_apply_draws")
        else:
            L.check(lib.ortk_mask_apply(L.ptr(self._flat), L.ptr(self._mask_flat), L.ptr(self._weff), self._n_train,
                                        self._mode(train), self._mask_seed(seed), L.stream_ptr()), "ortk_mask_apply")
        return L.ptr(self._weff)

    def set_mask_draws(self, draws):
        """Explicit uniforms for the training-mode Bernoulli sample (``None`` = the counter hash): a dict ``mask name -> array``
        with the mask's shape (or a flat arena-sized tensor).  The sample of a mask element is ``u < sigmoid(logit)``
        (pruning/sampler.py:10-17 for a given uniform), in the forward and in the straight-through backward."""
        if draws is None:
            self._mask_draws = None
            return
        if isinstance(draws, dict):
            flat = torch.zeros(self._n_train)
            for e in self.named_weight_entries():
                key = e["name"] + "_pruning_mask"
                if e["kind"] == 1 and key in draws:
                    flat[e["offset"]:e["offset"] + e["numel"]] = torch.as_tensor(draws[key], dtype=torch.float32).reshape(-1)
            draws = flat
        self._mask_draws = draws.to(self._flat.device).float().contiguous()

    def _draws(self, train):
        d = getattr(self, "_mask_draws", None)
        if d is None or not train or not self._supermask:
            return None
        if d.device != self._flat.device:
            d = self._mask_draws = d.to(self._flat.device)
        return d

    def _eff_params_tensor(self):
        return self._weff

    def _train_density(self, offset, N, K):
        """Supermask training samples Bernoulli(sigmoid(m)) (pruning/sampler.py:10-17): the expected density of a block is
        the mean of sigmoid(m), which is ABOVE the eval-mode round(sigmoid(m)) density the block selection sees; +15 % and
        four standard deviations of the sample on top (host sync, once per plan)."""
        if not self._supermask:
            return 0.0
        p = torch.sigmoid(self._mask_flat[offset: offset + N * K].float())
        mean = float(p.mean())
        return min(1.0, 1.15 * mean + 4.0 * (mean / max(N * K, 1)) ** 0.5)

    @staticmethod
    def _mask_seed(seed):
        return (int(seed) * 2654435761 + 0x5BD1E995) & 0xFFFFFFFF

    def _finish_grads(self, gflat, train, seed, sparsity_coef=None):
        """dW_eff -> (dW, dm): straight-through over the sample, real sigmoid derivative (sampler.py:10-66)."""
        lib = L.lib()
        if sparsity_coef is None:
            # set by compute_sparsity_loss() earlier in the same step (as in scripts/train_n_prune_transformer.py:
            # 143-149, loss = caption loss + sparsity loss, then ONE backward); consumed exactly once
            sparsity_coef, self._sparsity_coef = self._sparsity_coef, None
        need_dm = self._supermask or self.mask_type == prune.SNIP
        dm = torch.zeros(self._n_train, device=gflat.device) if need_dm else None
        draws = self._draws(train)
        if draws is not None:
            L.check(lib.ortk_mask_bwd_draws(L.ptr(gflat), L.ptr(self._flat), L.ptr(self._mask_flat), L.ptr(draws), L.ptr(gflat), L.ptr(dm),
                                            self._n_train, L.ptr(sparsity_coef), L.stream_ptr()), "ortk_mask_bwd_draws")
        else:
            L.check(lib.ortk_mask_bwd(L.ptr(gflat), L.ptr(self._flat), L.ptr(self._mask_flat), L.ptr(gflat), L.ptr(dm),
                                      self._n_train, self._mode(train), self._mask_seed(seed), L.ptr(sparsity_coef),
                                      L.stream_ptr()), "ortk_mask_bwd")
        out = []
        for e in self.named_weight_entries():
            sl = slice(e["offset"], e["offset"] + e["numel"])
            out.append(gflat[sl].view(e["shape"]))
            if e["kind"] == 1:
                out.append(dm[sl].view(e["shape"]) if need_dm else None)
        return out

    def _param_list(self):
        out = []
        for e in self.named_weight_entries():
            out.append(self._params[e["name"]][2])
            if e["kind"] == 1:
                out.append(self._params[e["name"] + "_pruning_mask"][2])
        return out

    def _active_mask_count(self):
        """(#active mask elements, device scalar of kept entries) with ONE launch over the arena when no scope is
        frozen: kept = count(arena) - #neutral positions."""
        lib = L.lib()
        cnt = torch.zeros(1, device=self._flat.device)
        if self.mask_freeze_scope is None:
            L.check(lib.ortk_mask_count(L.ptr(self._mask_flat), self._n_train, 0, L.ptr(cnt), L.stream_ptr()), "ortk_mask_count")
            return self.total_mask_params, cnt[0] - float(self._n_neutral)
        n = 0
        for _, m in self.active_pruning_masks():
            L.check(lib.ortk_mask_count(L.ptr(m.data), m.numel(), 0, L.ptr(cnt), L.stream_ptr()), "ortk_mask_count")
            n += m.numel()
        return n, cnt[0]

    @staticmethod
    def add_argparse_args(parser):
        _Dense.add_argparse_args(parser)
        PruningMixin.add_argparse_args(parser)
