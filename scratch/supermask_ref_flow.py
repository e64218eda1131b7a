"""What does the REFERENCE reach on the tiny ORT-prune model under the flow of tests/test_gpu_prune_flow.py (its own
tests/test_prune.py recipe: Adam lr 10 on the active masks, sparsity weight 120, 60 iterations, target 0.8)?"""
import sys, os
sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch
import common as C
from make_golden import import_reference, load_weights, tt
get_model, Config, losses, optim, prune = import_reference()
torch.manual_seed(8888)
cfg = Config(**dict(C.TINY_CFG, prune_type="supermask", prune_mask_freeze_scope="model.generator.", prune_supermask_init=5.0, drop_prob_src=0.1))
model = get_model("relation_transformer_prune")(cfg)
shapes = {n: tuple(p.shape) for n, p in model.named_parameters() if not n.endswith("_pruning_mask")}
sd = C.state_dict_from_shapes(shapes, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
with torch.no_grad():
    for n, p in model.named_parameters():
        if n in sd: p.copy_(torch.from_numpy(sd[n]))
tb = tt(C.make_inputs(**C.G1_INPUTS))
ITERS, TARGET = 60, 0.8
groups = [{"params": list(model.all_weights(named=False))},
          {"params": list(model.active_pruning_masks(named=False)), "lr": 10.0, "weight_decay": 0, "eps": 1e-8, "pruning_mask": True}]
opt = optim.get_optim(groups, Config(lr_scheduler="noam", optim="adam", d_model=cfg.d_model, noamopt_factor=0.1, noamopt_warmup=10))
crit = losses.LanguageModelCriterion()
model.train()
for i in range(ITERS):
    opt.zero_grad()
    logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = crit(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:]) + model.compute_sparsity_loss(TARGET, weight=120.0, current_step=i, max_step=ITERS)
    loss.backward()
    optim.clip_gradient(opt, 0.1)
    opt.step(epoch=0)
    if i % 10 == 9 or i < 3:
        print(i, "loss %.3f" % loss.item(), "active sparsity %.4f" % float(model.active_mask_sparsities[0]), "all %.4f" % float(model.all_mask_sparsities[0]))
print("final active sparsity", float(model.active_mask_sparsities[0]))
