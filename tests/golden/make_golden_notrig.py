#!/usr/bin/env python3
"""Golden fixture G5: tiny ORT with ``no_box_trigonometric_embedding=True`` (4-d geometry embedding,
relation_transformer.py:131-136,243-256), produced by running the REFERENCE on CPU.  Same recipe as G1
(weights / inputs from common.py).      python tests/golden/make_golden_notrig.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402
from make_golden import import_reference, load_weights, tt  # noqa: E402


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)
    get_model, Config, losses, optim, prune = import_reference()
    cfg = Config(**dict(C.TINY_CFG, no_box_trigonometric_embedding=True))
    model = get_model("relation_transformer")(cfg)
    load_weights(model, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
    model.eval()
    tb = tt(C.make_inputs(**C.G1_INPUTS))
    g = {}
    model.zero_grad()
    logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
    loss.backward()
    g["logp"] = logp.detach().numpy()
    g["xe_loss"] = np.float32(loss.item())
    for n, p in model.named_parameters():
        if ".WGs." in n or n.startswith("att_embed") or "encoder.layers.0.self_attn.linears.0" in n:
            g["grad/" + n] = p.grad.numpy().copy()
    g["grad_abs_sum"] = np.float64(sum(p.grad.double().abs().sum().item() for p in model.parameters()))
    with torch.no_grad():
        for bs in (1, 3):
            seq_o, lp_o = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                                opt={"beam_size": bs}, mode="sample")
            g[f"decode_b{bs}/seq"] = seq_o.numpy()
            g[f"decode_b{bs}/logprobs"] = lp_o.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_tiny_notrig.npz"), **g)
    print("g5: loss", float(g["xe_loss"]), "WG shape", tuple(model.model.encoder.layers[0].self_attn.WGs[0].weight.shape))


if __name__ == "__main__":
    main()
