#!/usr/bin/env python3
"""Golden G13: two supermask TRAINING steps of the reference with its two optimizer groups.

Follows scripts/train_n_prune_transformer.py:67-82,132-168 on the tiny ORT-prune model: group 0 = all weights under
Noam-Adam (betas 0.9/0.98, eps 1e-9), group 1 = the ACTIVE pruning masks with lr = prune_supermask_lr (untouched by Noam),
eps 1e-2, weight_decay 0; loss = XE + compute_sparsity_loss(target, weight, step, max_step); clip_grad_value_(0.1) on both
groups; ``prune_mask_freeze_scope = "model.generator."`` keeps the generator's mask logits out of group 1.  The model
runs in train() mode with dropout set to 0 and the Bernoulli draws INJECTED (u depends on the mask's shape and the step), so
the HIP path can replay the very same samples (ortk_mask_apply_draws).  Run in the build container only:
    python tests/golden/make_golden_maskopt.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402
from make_golden import import_reference, load_weights, tt  # noqa: E402

LR_MASK, TARGET, WEIGHT, MAX_STEP, STEPS = 100.0, 0.9, 30.0, 100, 2


def draws_for(shape, step):
    rs = np.random.RandomState((zlib.crc32(str(tuple(shape)).encode()) + 7919 * step) & 0x7FFFFFFF)
    return rs.uniform(size=tuple(shape)).astype(np.float32)


def main():
    get_model, Config, losses, optim, prune = import_reference()
    import torch
    cfg = Config(**dict(C.TINY_CFG, prune_type="supermask", prune_mask_freeze_scope="model.generator.", prune_supermask_init=5.0))
    model = get_model("relation_transformer_prune")(cfg)
    load_weights(model, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, keep_prob=C.G3_KEEP)
    model.train()
    for m_ in model.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    tb = tt(C.make_inputs(**C.G1_INPUTS))
    groups = [{"params": list(model.all_weights(named=False))},
              {"params": list(model.active_pruning_masks(named=False)), "lr": LR_MASK, "weight_decay": 0, "eps": 1e-2, "pruning_mask": True}]
    opt = optim.get_optim(groups, Config(lr_scheduler="noam", optim="adam", d_model=cfg.d_model, noamopt_factor=1.0, noamopt_warmup=10))
    crit = losses.LanguageModelCriterion()
    g = {"losses": [], "caption_losses": []}
    orig = torch.bernoulli
    try:
        for step in range(STEPS):
            torch.bernoulli = lambda p, *a, _s=step, **k: (torch.from_numpy(draws_for(p.shape, _s)) < p).to(p.dtype)
            opt.zero_grad()
            logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
            loss = crit(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
            g["caption_losses"].append(loss.item())
            loss = loss + model.compute_sparsity_loss(TARGET, weight=WEIGHT, current_step=step, max_step=MAX_STEP)
            loss.backward()
            optim.clip_gradient(opt, 0.1)
            opt.step(epoch=0)
            g["losses"].append(loss.item())
    finally:
        torch.bernoulli = orig
    out = {"losses": np.array(g["losses"], np.float32), "caption_losses": np.array(g["caption_losses"], np.float32),
           "meta": np.array([LR_MASK, TARGET, WEIGHT, MAX_STEP, STEPS], np.float64)}
    for n, p in model.named_parameters():
        if n.endswith("_pruning_mask") and any(s in n for s in ("att_embed.0.weight", "encoder.layers.0.self_attn.linears.0.weight",
                                                                 "decoder.layers.1.feed_forward.w_1.weight", "decoder.layers.0.src_attn.linears.1.weight",
                                                                 "generator.proj.weight", "tgt_embed.0.lut.weight")):
            out["mask/" + n] = p.detach().numpy().copy()
        if n in ("att_embed.0.weight", "model.decoder.layers.1.feed_forward.w_1.weight", "model.generator.proj.bias"):
            out["param/" + n] = p.detach().numpy().copy()
    out["mask_abs_sum"] = np.float64(sum(p.detach().double().abs().sum().item() for n, p in model.all_pruning_masks()))
    np.savez_compressed(os.path.join(HERE, "g13_tiny_maskopt.npz"), **out)
    print("g13: losses", g["losses"], "caption", g["caption_losses"], "mask_abs_sum", out["mask_abs_sum"])
    gm = out["mask/model.generator.proj.weight_pruning_mask"]
    print("frozen generator logits unchanged:", float(gm.min()), float(gm.max()))


if __name__ == "__main__":
    main()
