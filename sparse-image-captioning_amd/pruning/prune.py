"""Pruning API of the reference's ``PruningMixin`` (``sparse_caption/pruning/prune.py:46-433``) for the arena model.

Per-step work (masked-weight materialisation, straight-through backward, kept-entry counts) runs in HIP
(``ortk_mask_apply`` / ``ortk_mask_bwd`` / ``ortk_mask_count``).  The occasional host-driven mask updates
(one-shot / gradual magnitude pruning, SNIP) use torch tensor ops on the device arenas: they are off the
per-step path and are plumbing around ``torch.topk``.
"""
import math

import torch

MASK_FREEZE = "mask_freeze"
REGULAR = "supermask"
MAG_BLIND, MAG_UNIFORM, MAG_DIST = "mag_blind", "mag_uniform", "mag_dist"
MAG_GRAD_BLIND, MAG_GRAD_UNIFORM, MAG_GRAD_DIST = "mag_grad_blind", "mag_grad_uniform", "mag_grad_dist"
LOTTERY_MAG_BLIND, LOTTERY_MAG_UNIFORM, LOTTERY_MAG_DIST = "lottery_mag_blind", "lottery_mag_uniform", "lottery_mag_dist"
LOTTERY_MASK_FREEZE = "lottery_mask_freeze"
SNIP = "snip"
SUPER_MASKS = [REGULAR]
MAG_ANNEAL = [MAG_GRAD_BLIND, MAG_GRAD_UNIFORM]
MAG_HARD = [MAG_BLIND, MAG_UNIFORM, MAG_DIST]
LOTTERY = [LOTTERY_MAG_BLIND, LOTTERY_MAG_UNIFORM, LOTTERY_MAG_DIST, LOTTERY_MASK_FREEZE]
MAG_PRUNE_MASKS = MAG_HARD + MAG_ANNEAL + LOTTERY + [SNIP]
VALID_MASKS = SUPER_MASKS + MAG_PRUNE_MASKS + [MASK_FREEZE]


def rounding_sigmoid(m):
    return torch.round(torch.sigmoid(m))


class PruningMixin:
    """Same method / property names as the reference mixin; `self` is an nn.Module whose masked weights have a
    sibling parameter ``<name>_pruning_mask``."""

    def _init_pruning(self, mask_type, mask_freeze_scope=""):
        assert mask_type in VALID_MASKS, f"`mask_type` must be one of {VALID_MASKS}, saw `{mask_type}`"
        assert isinstance(mask_freeze_scope, str)
        self.mask_type = mask_type
        self.mask_freeze_scope = None if mask_freeze_scope == "" else [_ for _ in mask_freeze_scope.split(",") if _ != ""]
        self.sparsity_target = 0.0
        self.sparsity_loss = {}

    # --- enumerators (prune.py:67-114)
    def all_pruning_masks(self, named=True):
        return [(n, p) if named else p for n, p in self.named_parameters() if n.endswith("_pruning_mask")]

    def all_pruned_weights(self, named=True):
        names = set(n.replace("_pruning_mask", "") for n, _ in self.all_pruning_masks())
        return [(n, p) if named else p for n, p in self.named_parameters() if n in names]

    def all_weights(self, named=True):
        return [(n, p) if named else p for n, p in self.named_parameters() if not n.endswith("_pruning_mask")]

    def active_pruning_masks(self, named=True):
        if self.mask_freeze_scope is None:
            return self.all_pruning_masks(named)
        return [(n, p) if named else p for n, p in self.all_pruning_masks()
                if not any(n.startswith(_) for _ in self.mask_freeze_scope)]

    def active_pruned_weights(self, named=True):
        names = set(n.replace("_pruning_mask", "") for n, _ in self.active_pruning_masks())
        return [(n, p) if named else p for n, p in self.named_parameters() if n in names]

    def trainable_pruning_masks(self, named=True):
        return [(n, p) if named else p for n, p in self.all_pruning_masks() if p.requires_grad]

    @property
    def total_mask_params(self):
        return sum(_.nelement() for _ in self.all_pruning_masks(named=False))

    @property
    def total_weight_params(self):
        return sum(_.nelement() for _ in self.all_weights(named=False))

    # --- statistics (prune.py:124-163)
    @staticmethod
    def calculate_sparsities(tensor_list, count_nnz_fn):
        nelem = [_.nelement() for _ in tensor_list]
        nnz = [count_nnz_fn(_) for _ in tensor_list]
        sps = [1.0 - (z / n) for z, n in zip(nnz, nelem)]
        total_nnz = sum(nnz)
        return 1.0 - (total_nnz / sum(nelem)), total_nnz, sps

    def _binarised(self, masks):
        return [rounding_sigmoid(_) for _ in masks] if self.mask_type in SUPER_MASKS else list(masks)

    @property
    def all_weight_sparsities(self):
        names, weights = zip(*self.all_pruned_weights(named=True))
        return self.calculate_sparsities(weights, lambda t: t.ne(0).float().sum()) + (names,)

    @property
    @torch.no_grad()
    def all_mask_sparsities(self):
        names, masks = zip(*self.all_pruning_masks(named=True))
        return self.calculate_sparsities(self._binarised(masks), torch.sum) + (names,)

    @property
    @torch.no_grad()
    def active_mask_sparsities(self):
        names, masks = zip(*self.active_pruning_masks(named=True))
        return self.calculate_sparsities(self._binarised(masks), torch.sum) + (names,)

    @property
    def all_mask_avg(self):
        return torch.cat([m.detach().reshape(-1) for m in self.all_pruning_masks(named=False)]).mean()

    @property
    def active_mask_avg(self):
        return torch.cat([m.detach().reshape(-1) for m in self.active_pruning_masks(named=False)]).mean()

    @torch.no_grad()
    def prune_weights(self):
        """w[:] = w * mask (prune.py:165-174)."""
        masks = self._binarised(self.all_pruning_masks(named=False))
        for w, m in zip(self.all_pruned_weights(named=False), masks):
            w.mul_(m)

    # --- checkpoints (prune.py:176-226)
    def state_dict_dense(self, destination=None, prefix="", keep_vars=False, discard_pruning_mask=False,
                         prune_weights=True, binarize_supermasks=False):
        if prune_weights:
            self.prune_weights()
        sd = self.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)
        if discard_pruning_mask and binarize_supermasks:
            raise ValueError("`discard_pruning_mask` and `binarize_supermasks` cannot be True at the same time.")
        if discard_pruning_mask:
            for n, _ in self.all_pruning_masks():
                del sd[prefix + n]
        if binarize_supermasks:
            if self.mask_type not in SUPER_MASKS:
                raise ValueError(f"`binarize_supermasks` can only be True for mask_type in {SUPER_MASKS}.")
            for n, _ in self.all_pruning_masks():
                sd[prefix + n] = rounding_sigmoid(sd[prefix + n])
        return sd

    def state_dict_sparse(self, destination=None, prefix="", keep_vars=False, discard_pruning_mask=True,
                          prune_weights=True, binarize_supermasks=False):
        sd = self.state_dict_dense(destination, prefix, keep_vars, discard_pruning_mask, prune_weights, binarize_supermasks)
        pruned = set(n for n, _ in self.all_pruned_weights(named=True))
        return {k: v.detach().clone().to_sparse() if (isinstance(v, torch.Tensor) and k in pruned) else v.detach().clone()
                for k, v in sd.items()}

    def load_sparse_state_dict(self, sparse_state_dict, strict=True):
        self.load_state_dict({k: v.to_dense() if v.is_sparse else v for k, v in sparse_state_dict.items()}, strict=strict)

    # --- supermask sparsity loss (prune.py:228-269)
    def compute_sparsity_loss(self, sparsity_target, weight, current_step, max_step):
        assert self.mask_type in SUPER_MASKS, f"Invalid mask type. Must be one of {SUPER_MASKS}"
        n_active, kept = self._active_mask_count()           # device scalar from ortk_mask_count
        if n_active == 0:
            return 0.0
        total_sparsity = 1.0 - kept / n_active
        loss = torch.abs(sparsity_target - total_sparsity)
        self.sparsity_loss = {"loss": loss}
        step = 1.0 + math.cos(min(1.0, current_step / max_step) * math.pi)
        anneal_rate = step / 2
        scaled = loss * weight * (1.0 - anneal_rate)
        self.sparsity_loss["anneal_rate"] = anneal_rate
        self.sparsity_loss["loss_scaled"] = scaled
        # d(scaled)/d(sample) for every active mask element, consumed by ortk_mask_bwd (straight-through Round)
        self._sparsity_coef = (torch.sign(sparsity_target - total_sparsity) * (weight * (1.0 - anneal_rate) / n_active)).reshape(1).float()
        return scaled

    # --- magnitude / SNIP pruning (prune.py:271-433)
    @staticmethod
    def compute_mask(criterion, sparsity_target):
        assert isinstance(sparsity_target, float) and 0 <= sparsity_target < 1.0
        mask = torch.ones_like(criterion)
        k = int(sparsity_target * criterion.nelement())
        assert 0 <= k < criterion.nelement()
        if k > 0:
            idx = torch.topk(criterion.reshape(-1), k=k, largest=False).indices
            mask.view(-1)[idx] = 0
        return mask

    @torch.no_grad()
    def sparsity_check(self, warning_threshold=0.999):
        _, _, sps, names = self.all_mask_sparsities
        return [(n, float(s)) for n, s in zip(names, sps) if float(s) > warning_threshold]

    @torch.no_grad()
    def update_masks_once(self, sparsity_target):
        assert self.mask_type in MAG_PRUNE_MASKS, f"Invalid mask_type: {self.mask_type}. Must be one of {MAG_PRUNE_MASKS}"
        _, masks = zip(*self.active_pruning_masks())
        _, weights = zip(*self.active_pruned_weights())
        assert len(weights) == len(masks)
        if self.mask_type == SNIP:
            saliency = [_.grad for _ in masks]
            assert all(_ is not None for _ in saliency)
            vec = torch.cat([s.reshape(-1) for s in saliency])
            criterion = [vec / vec.sum()]
        elif self.mask_type in (MAG_DIST, MAG_GRAD_DIST, LOTTERY_MAG_DIST):
            cs = []
            for w in weights:
                sd = torch.std(w.reshape(-1), dim=0, unbiased=False)
                cs.append(torch.abs((w - w.mean()) / sd).reshape(-1))
            criterion = [torch.cat(cs)]
        elif self.mask_type in (MAG_UNIFORM, MAG_GRAD_UNIFORM, LOTTERY_MAG_UNIFORM):
            criterion = [torch.abs(w) for w in weights]
        elif self.mask_type in (MAG_BLIND, MAG_GRAD_BLIND, LOTTERY_MAG_BLIND):
            criterion = [torch.cat([torch.abs(w).reshape(-1) for w in weights])]
        else:
            raise ValueError(f"Unknown `self.mask_type`: {self.mask_type}")
        new_masks = [self.compute_mask(c, sparsity_target) for c in criterion]
        if len(new_masks) == 1:
            new_masks = torch.split(new_masks[0], [m.nelement() for m in masks])
        assert len(new_masks) == len(masks)
        for m, nm in zip(masks, new_masks):
            m.data.view(-1)[:] = nm.reshape(-1)
        self.sparsity_target = sparsity_target
        self.sparsity_check()
        return True

    @torch.no_grad()
    def update_masks_gradual(self, sparsity_target, current_step, start_step, prune_steps, initial_sparsity=0.0,
                             prune_frequency=1000):
        t, si, sf, t0, dt = current_step, initial_sparsity, sparsity_target, start_step, prune_frequency
        tn = start_step + prune_frequency * prune_steps
        assert self.mask_type in MAG_ANNEAL
        assert dt > 0 and prune_steps > 0 and (tn - t0) % dt == 0
        if (t >= t0) and ((t <= tn) or (tn < 0)) and ((t - t0) % dt) == 0:
            p = min(1.0, max(0.0, (t - t0) / (tn - t0)))
            self.update_masks_once(sparsity_target=sf + ((si - sf) * ((1.0 - p) ** 3)))
        return False

    @staticmethod
    def add_argparse_args(parser):
        g = parser.add_argument_group("Pruning", "Arguments for weight pruning.")
        g.add_argument("--prune_type", type=str, default="", choices=VALID_MASKS)
        g.add_argument("--prune_sparsity_target", type=float, default=0.8)
        g.add_argument("--prune_mask_freeze_scope", type=str, default="")
        g.add_argument("--prune_snip_grad_accum", type=int, default=1)
        g.add_argument("--prune_supermask_init", type=float, default=5.0)
        g.add_argument("--prune_supermask_sparsity_weight", type=float, default=-1.0)
        g.add_argument("--prune_supermask_lr", type=float, default=1e2)
        g.add_argument("--prune_supermask_bypass_sigmoid_grad", action="store_true")
