#!/usr/bin/env python3
"""Golden fixture G10: the reference's plain `transformer` model (sparse_caption/models/transformer.py:617-719) on the tiny
configuration, produced by running the REFERENCE on CPU.  Same recipe as G1 (weights from the per-name seeded generator,
inputs of G1 without the boxes).
    python tests/golden/make_golden_plain.py      # writes tests/golden/g10_tiny_plain_transformer.npz
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
    cfg = Config(**C.TINY_CFG)
    model = get_model("transformer")(cfg)
    load_weights(model, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
    model.eval()
    tb = tt(C.make_inputs(**C.G1_INPUTS))
    g = {}
    g["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
    g["param_names"] = np.array([n for n, _ in model.named_parameters()])
    g["param_shapes"] = np.array([",".join(str(int(x)) for x in p.shape) for _, p in model.named_parameters()])
    g["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    model.zero_grad()
    logp = model(att_feats=tb["att_feats"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
    loss.backward()
    g["logp"] = logp.detach().numpy()
    g["xe_loss"] = np.float32(loss.item())
    for n, p in model.named_parameters():
        g["grad/" + n] = p.grad.numpy().copy()
    with torch.no_grad():
        memory, _ = model.core.encode(src=tb["att_feats"], src_mask=tb["att_masks"])
        g["memory"] = memory.numpy()
        for bs in (1, 3):
            seq_o, lp_o = model(att_feats=tb["att_feats"], att_masks=tb["att_masks"], opt={"beam_size": bs}, mode="sample")
            g[f"decode_b{bs}/seq"] = seq_o.numpy()
            g[f"decode_b{bs}/logprobs"] = lp_o.numpy()
    np.savez_compressed(os.path.join(HERE, "g10_tiny_plain_transformer.npz"), **g)
    print("g10: loss", float(g["xe_loss"]), "params", int(g["n_params"]), "keys", len(g["state_dict_keys"]))


if __name__ == "__main__":
    main()
