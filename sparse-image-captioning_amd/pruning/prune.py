"""Pruning interface of the arena model: the public surface of the reference's ``PruningMixin``
(``sparse_caption/pruning/prune.py:46-433``: same attribute, property, method and option names, same return shapes) so
that ``train_n_prune_transformer.py`` and ``eval_model.py`` drive it unchanged.

Where the work happens here:

* every training step — materialising ``w * mask``, the straight-through backward, counting kept entries — is HIP
  (``ortk_mask_apply`` / ``ortk_mask_bwd`` / ``ortk_mask_count`` over the flat arenas, see ``relation_transformer_prune.py``);
* the occasional mask UPDATES (one-shot / gradual magnitude pruning, SNIP), the statistics and the checkpoint views are a few
  tensor expressions on the device arenas below: off the per-step path.

Mask kinds are described by one table (``_KIND``) instead of the reference's parallel lists; the module-level names the
callers use (``REGULAR``, ``SNIP``, ``MAG_HARD`` ...) are derived from it.
"""
import math

import torch

_SUFFIX = "_pruning_mask"

# kind -> (family, criterion).  family: "super" = trainable logits binarised by round(sigmoid), "hard" = one-shot magnitude,
# "anneal" = gradual magnitude, "lottery" = lottery-ticket variants, "snip", "freeze".  criterion: how weights are ranked.
_KIND = {
    "supermask": ("super", None),
    "mag_blind": ("hard", "blind"), "mag_uniform": ("hard", "uniform"), "mag_dist": ("hard", "dist"),
    "mag_grad_blind": ("anneal", "blind"), "mag_grad_uniform": ("anneal", "uniform"), "mag_grad_dist": (None, "dist"),
    "lottery_mag_blind": ("lottery", "blind"), "lottery_mag_uniform": ("lottery", "uniform"), "lottery_mag_dist": ("lottery", "dist"),
    "lottery_mask_freeze": ("lottery", None),
    "snip": ("snip", "snip"),
    "mask_freeze": ("freeze", None),
}


def _family(*families):
    return [k for k, (f, _) in _KIND.items() if f in families]


MASK_FREEZE, REGULAR, SNIP = "mask_freeze", "supermask", "snip"
MAG_BLIND, MAG_UNIFORM, MAG_DIST = "mag_blind", "mag_uniform", "mag_dist"
MAG_GRAD_BLIND, MAG_GRAD_UNIFORM, MAG_GRAD_DIST = "mag_grad_blind", "mag_grad_uniform", "mag_grad_dist"
LOTTERY_MAG_BLIND, LOTTERY_MAG_UNIFORM, LOTTERY_MAG_DIST = "lottery_mag_blind", "lottery_mag_uniform", "lottery_mag_dist"
LOTTERY_MASK_FREEZE = "lottery_mask_freeze"
SUPER_MASKS = _family("super")
MAG_HARD = _family("hard")
MAG_ANNEAL = _family("anneal")                 # (the reference lists mag_grad_dist nowhere: it is not a valid choice there either)
LOTTERY = _family("lottery")
MAG_PRUNE_MASKS = MAG_HARD + MAG_ANNEAL + LOTTERY + [SNIP]
VALID_MASKS = SUPER_MASKS + MAG_PRUNE_MASKS + [MASK_FREEZE]


def rounding_sigmoid(m):
    """Binarisation of supermask logits (masked_layer.py: round(sigmoid(m)))."""
    return torch.sigmoid(m).round()


def _rank_by(kind, weights, masks):
    """Pruning criterion tensors for one update: a single flat tensor (global ranking) or one tensor per weight (per-layer)."""
    how = _KIND[kind][1]
    if how == "snip":                           # connection sensitivity: |dL/dmask|, normalised over the whole model
        grads = [m.grad for m in masks]
        if any(g is None for g in grads):
            raise AssertionError("SNIP needs the gradient of every active mask (run a backward pass first)")
        flat = torch.cat([g.reshape(-1) for g in grads])
        return [flat / flat.sum()]
    if how == "dist":                           # |z-score| inside each layer, ranked globally
        z = [((w - w.mean()) / w.reshape(-1).std(unbiased=False)).abs().reshape(-1) for w in weights]
        return [torch.cat(z)]
    if how == "uniform":                        # |w| ranked inside each layer
        return [w.abs() for w in weights]
    if how == "blind":                          # |w| ranked over the whole model
        return [torch.cat([w.abs().reshape(-1) for w in weights])]
    raise ValueError(f"Unknown `self.mask_type`: {kind}")


class PruningMixin:
    """Mixed into an ``nn.Module`` whose maskable weights each have a sibling parameter ``<name>_pruning_mask``."""

    def _init_pruning(self, mask_type, mask_freeze_scope=""):
        assert mask_type in VALID_MASKS, f"`mask_type` must be one of {VALID_MASKS}, saw `{mask_type}`"
        assert isinstance(mask_freeze_scope, str)
        scopes = [s for s in mask_freeze_scope.split(",") if s]
        self.mask_type = mask_type
        self.mask_freeze_scope = scopes or None
        self.sparsity_target = 0.0
        self.sparsity_loss = {}

    # ------------------------------------------------------------------ parameter views
    def _pick(self, keep, named):
        out = [(n, p) for n, p in self.named_parameters() if keep(n, p)]
        return out if named else [p for _, p in out]

    def _frozen(self, mask_name):
        return self.mask_freeze_scope is not None and any(mask_name.startswith(s) for s in self.mask_freeze_scope)

    def all_pruning_masks(self, named=True):
        return self._pick(lambda n, p: n.endswith(_SUFFIX), named)

    def active_pruning_masks(self, named=True):
        return self._pick(lambda n, p: n.endswith(_SUFFIX) and not self._frozen(n), named)

    def trainable_pruning_masks(self, named=True):
        return self._pick(lambda n, p: n.endswith(_SUFFIX) and p.requires_grad, named)

    def all_weights(self, named=True):
        return self._pick(lambda n, p: not n.endswith(_SUFFIX), named)

    def _weights_of(self, mask_items, named):
        owners = {n[:-len(_SUFFIX)] for n, _ in mask_items}
        return self._pick(lambda n, p: n in owners, named)

    def all_pruned_weights(self, named=True):
        return self._weights_of(self.all_pruning_masks(), named)

    def active_pruned_weights(self, named=True):
        return self._weights_of(self.active_pruning_masks(), named)

    @property
    def total_mask_params(self):
        return sum(m.nelement() for m in self.all_pruning_masks(named=False))

    @property
    def total_weight_params(self):
        return sum(w.nelement() for w in self.all_weights(named=False))

    # ------------------------------------------------------------------ statistics
    @staticmethod
    def calculate_sparsities(tensor_list, count_nnz_fn):
        """-> (overall sparsity, total non-zeros, per-tensor sparsities)."""
        sizes = [t.nelement() for t in tensor_list]
        kept = [count_nnz_fn(t) for t in tensor_list]
        total = sum(kept)
        return 1.0 - (total / sum(sizes)), total, [1.0 - (k / n) for k, n in zip(kept, sizes)]

    def _binarised(self, masks):
        return [rounding_sigmoid(m) for m in masks] if self.mask_type in SUPER_MASKS else list(masks)

    def _mask_stats(self, items):
        names, masks = zip(*items)
        return self.calculate_sparsities(self._binarised(masks), torch.sum) + (names,)

    @property
    def all_weight_sparsities(self):
        names, weights = zip(*self.all_pruned_weights(named=True))
        return self.calculate_sparsities(weights, lambda t: t.ne(0).float().sum()) + (names,)

    @property
    @torch.no_grad()
    def all_mask_sparsities(self):
        return self._mask_stats(self.all_pruning_masks(named=True))

    @property
    @torch.no_grad()
    def active_mask_sparsities(self):
        return self._mask_stats(self.active_pruning_masks(named=True))

    @staticmethod
    def _mean_of(masks):
        return torch.cat([m.detach().reshape(-1) for m in masks]).mean()

    @property
    def all_mask_avg(self):
        return self._mean_of(self.all_pruning_masks(named=False))

    @property
    def active_mask_avg(self):
        return self._mean_of(self.active_pruning_masks(named=False))

    @torch.no_grad()
    def sparsity_check(self, warning_threshold=0.999):
        """Layers that are (almost) entirely pruned: [(name, sparsity)]."""
        _, _, per_layer, names = self.all_mask_sparsities
        return [(n, float(s)) for n, s in zip(names, per_layer) if float(s) > warning_threshold]

    # ------------------------------------------------------------------ checkpoints
    @torch.no_grad()
    def prune_weights(self):
        """Bake the masks into the weights in place."""
        for w, m in zip(self.all_pruned_weights(named=False), self._binarised(self.all_pruning_masks(named=False))):
            w.mul_(m)

    def state_dict_dense(self, destination=None, prefix="", keep_vars=False, discard_pruning_mask=False,
                         prune_weights=True, binarize_supermasks=False):
        if discard_pruning_mask and binarize_supermasks:
            raise ValueError("`discard_pruning_mask` and `binarize_supermasks` cannot be True at the same time.")
        if binarize_supermasks and self.mask_type not in SUPER_MASKS:
            raise ValueError(f"`binarize_supermasks` can only be True for mask_type in {SUPER_MASKS}.")
        if prune_weights:
            self.prune_weights()
        sd = self.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)
        for n, _ in self.all_pruning_masks():
            if discard_pruning_mask:
                del sd[prefix + n]
            elif binarize_supermasks:
                sd[prefix + n] = rounding_sigmoid(sd[prefix + n])
        return sd

    def state_dict_sparse(self, destination=None, prefix="", keep_vars=False, discard_pruning_mask=True,
                          prune_weights=True, binarize_supermasks=False):
        """Dense state dict with the pruned weights as COO tensors."""
        dense = self.state_dict_dense(destination, prefix, keep_vars, discard_pruning_mask, prune_weights, binarize_supermasks)
        coo = {n for n, _ in self.all_pruned_weights(named=True)}
        out = {}
        for k, v in dense.items():
            v = v.detach().clone()
            out[k] = v.to_sparse() if (isinstance(v, torch.Tensor) and k in coo) else v
        return out

    def load_sparse_state_dict(self, sparse_state_dict, strict=True):
        self.load_state_dict({k: (v.to_dense() if v.is_sparse else v) for k, v in sparse_state_dict.items()}, strict=strict)

    # ------------------------------------------------------------------ supermask sparsity loss
    def compute_sparsity_loss(self, sparsity_target, weight, current_step, max_step):
        """|target - sparsity| * weight * (1 - cosine anneal), prune.py:228-269.  The kept-entry count comes from the device
        (``_active_mask_count`` -> ``ortk_mask_count``); its gradient w.r.t. every active mask sample is the constant stored in
        ``_sparsity_coef``, which ``ortk_mask_bwd`` adds (straight-through Round)."""
        assert self.mask_type in SUPER_MASKS, f"Invalid mask type. Must be one of {SUPER_MASKS}"
        n_active, kept = self._active_mask_count()
        if n_active == 0:
            return 0.0
        gap = sparsity_target - (1.0 - kept / n_active)
        anneal_rate = (1.0 + math.cos(min(1.0, current_step / max_step) * math.pi)) / 2
        gain = weight * (1.0 - anneal_rate)
        loss = torch.abs(gap)
        self.sparsity_loss = {"loss": loss, "anneal_rate": anneal_rate, "loss_scaled": loss * gain}
        self._sparsity_coef = (torch.sign(gap) * (gain / n_active)).reshape(1).float()
        return self.sparsity_loss["loss_scaled"]

    # ------------------------------------------------------------------ magnitude / SNIP mask updates
    @staticmethod
    def compute_mask(criterion, sparsity_target):
        """0/1 mask that drops the `sparsity_target` fraction of entries with the smallest criterion."""
        assert isinstance(sparsity_target, float) and 0 <= sparsity_target < 1.0
        n_drop = int(sparsity_target * criterion.nelement())
        assert 0 <= n_drop < criterion.nelement()
        mask = torch.ones_like(criterion)
        if n_drop:
            mask.view(-1)[torch.topk(criterion.reshape(-1), k=n_drop, largest=False).indices] = 0
        return mask

    @torch.no_grad()
    def update_masks_once(self, sparsity_target):
        assert self.mask_type in MAG_PRUNE_MASKS, f"Invalid mask_type: {self.mask_type}. Must be one of {MAG_PRUNE_MASKS}"
        masks = self.active_pruning_masks(named=False)
        weights = self.active_pruned_weights(named=False)
        assert len(weights) == len(masks)
        fresh = [self.compute_mask(c, sparsity_target) for c in _rank_by(self.mask_type, weights, masks)]
        if len(fresh) == 1:                     # global ranking: cut the flat mask back into layers
            fresh = torch.split(fresh[0], [m.nelement() for m in masks])
        assert len(fresh) == len(masks)
        for m, f in zip(masks, fresh):
            m.data.view(-1).copy_(f.reshape(-1))
        self.sparsity_target = sparsity_target
        self.sparsity_check()
        return True

    @torch.no_grad()
    def update_masks_gradual(self, sparsity_target, current_step, start_step, prune_steps, initial_sparsity=0.0,
                             prune_frequency=1000):
        """Cubic schedule of Zhu & Gupta, applied every `prune_frequency` steps from `start_step` (prune.py:383-433)."""
        assert self.mask_type in MAG_ANNEAL
        end_step = start_step + prune_frequency * prune_steps
        assert prune_frequency > 0 and prune_steps > 0 and (end_step - start_step) % prune_frequency == 0
        due = current_step >= start_step and (current_step <= end_step or end_step < 0)
        if due and (current_step - start_step) % prune_frequency == 0:
            done = min(1.0, max(0.0, (current_step - start_step) / (end_step - start_step)))
            self.update_masks_once(sparsity_target=sparsity_target + (initial_sparsity - sparsity_target) * (1.0 - done) ** 3)
        return False

    @staticmethod
    def add_argparse_args(parser):
        g = parser.add_argument_group("Pruning", "Arguments for weight pruning.")
        for flag, kw in (("--prune_type", dict(type=str, default="", choices=VALID_MASKS)),
                         ("--prune_sparsity_target", dict(type=float, default=0.8)),
                         ("--prune_mask_freeze_scope", dict(type=str, default="")),
                         ("--prune_snip_grad_accum", dict(type=int, default=1)),
                         ("--prune_supermask_init", dict(type=float, default=5.0)),
                         ("--prune_supermask_sparsity_weight", dict(type=float, default=-1.0)),
                         ("--prune_supermask_lr", dict(type=float, default=1e2)),
                         ("--prune_supermask_bypass_sigmoid_grad", dict(action="store_true"))):
            g.add_argument(flag, **kw)
