"""The reference's own pruning test (tests/test_prune.py: TestPrune._test_model) replayed on the HIP ORT-prune model
with the native trainer instead of its toy Embedding/LSTM/Linear model: for every mask type — initial sparsity 0,
one-shot pruning hits the target within 0.05, a short training run ends at the target within 0.05 for the magnitude / SNIP /
lottery masks and — for the supermask — where the reference's own run of this recipe on this model ends (0.482 +- 0.05), the frozen scope keeps the active sparsity above the overall one, weights
are only zeroed by `prune_weights()`."""
import pytest
import torch

import common as C
import helpers as H

pytestmark = pytest.mark.gpu

SPARSITY_TARGET = 0.8
ITERS = 60      # the reference test runs 40 iterations of its 60-parameter toy model


@pytest.fixture(scope="module")
def P():
    import sparse_image_captioning_amd as pkg
    pkg._lib.require_gpu()
    return pkg


@pytest.mark.parametrize("mask_type", ["supermask", "mag_blind", "mag_dist", "mag_uniform", "snip", "mag_grad_blind",
                                       "mag_grad_uniform", "lottery_mag_blind", "lottery_mag_uniform", "lottery_mag_dist"])
def test_prune_flow_like_reference_test(P, mask_type):
    from sparse_image_captioning_amd.pruning import prune
    from sparse_image_captioning_amd.training import NativeTrainer
    from sparse_image_captioning_amd.utils.config import Config
    from sparse_image_captioning_amd.utils.losses import LanguageModelCriterion
    torch.manual_seed(8888)
    cfg = Config(**dict(C.TINY_CFG, prune_type=mask_type, prune_mask_freeze_scope="model.generator.", prune_supermask_init=5.0,
                        drop_prob_src=0.1))
    model = P.get_model("relation_transformer_prune")(cfg)
    shapes = {k: v for k, v in H.prune_param_shapes(C.TINY_CFG).items() if not k.endswith("_pruning_mask")}
    model.load_state_dict(H.torch_state(shapes, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS), strict=False)
    model = model.cuda()
    b = {k: v.cuda() for k, v in H.g1_batch().items()}
    sparsity = lambda active: float((model.active_mask_sparsities if active else model.all_mask_sparsities)[0])
    assert sparsity(False) == 0, "Initial sparsity should be zero"

    if mask_type in prune.MAG_HARD + prune.LOTTERY + [prune.SNIP]:
        if mask_type == prune.SNIP:
            model.eval()
            logp = model(att_feats=b["att_feats"], boxes=b["boxes"], seqs=b["seqs"], att_masks=b["att_masks"])
            LanguageModelCriterion()(logp, b["seqs"][:, 1:], b["masks"][:, 1:]).backward()
        model.update_masks_once(sparsity_target=SPARSITY_TARGET)
        assert abs(sparsity(True) - SPARSITY_TARGET) < 0.05, "one-shot pruning sparsity"

    model.train()
    # the reference test drives the masks with torch.optim.Adam(lr=10) at its default eps (tests/test_prune.py:52-63)
    tr = NativeTrainer(model, noamopt_factor=0.1, noamopt_warmup=10, prune_supermask_lr=10.0, mask_eps=1e-8,
                       sparsity_target=SPARSITY_TARGET if mask_type == prune.REGULAR else None, sparsity_weight=120.0,
                       max_train_step=ITERS)
    for i in range(ITERS):
        loss = tr.xe_step(b)
        assert torch.isfinite(loss).all()
        if mask_type in prune.MAG_ANNEAL:
            model.update_masks_gradual(sparsity_target=SPARSITY_TARGET, current_step=i, start_step=10,
                                       prune_steps=max(1, ITERS // 4), prune_frequency=3)
    # Supermask: the reference's test allows 0.3 on its 60-parameter toy model.  On THIS model the reference itself (same
    # recipe: Adam lr 10 on the active masks, weight 120, 60 iterations, frozen generator) ends at an active sparsity of
    # 0.482 — the first lr-10 Adam step sends every logit from 5 to about -5 or 15, where sigma' ~ 0, and nothing moves
    # afterwards (scratch/supermask_ref_flow.py runs the reference; printed trace: 0.485, 0.4625, 0.482, ... 0.4819).  The
    # HIP path must land where the reference lands, not merely inside a wide band around the target.
    REF_FINAL = 0.482
    if mask_type == prune.REGULAR:
        assert abs(sparsity(True) - REF_FINAL) < 0.05, ("final sparsity vs the reference's own run", sparsity(True))
    else:
        assert abs(sparsity(True) - SPARSITY_TARGET) < 0.05, "final sparsity"
    assert sparsity(True) > sparsity(False), "active sparsity is higher: the generator is not pruned"
    assert float(model.all_weight_sparsities[0]) == 0, "weights are not pruned yet"
    model.prune_weights()
    assert abs(float(model.all_weight_sparsities[0]) - SPARSITY_TARGET) < 0.35, "weight sparsity after prune_weights()"
