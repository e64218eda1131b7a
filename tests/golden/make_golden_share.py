#!/usr/bin/env python3
"""Golden fixture G8: ACORT layer sharing (`share_layer_encoder` / `share_layer_decoder`, relation_transformer.py:80-88,
transformer.py:175-183) on a 3-layer tiny ORT, produced by running the REFERENCE on CPU.  Same recipe as G1.
    python tests/golden/make_golden_share.py      # writes tests/golden/g8_tiny_share_layer.npz
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402
from make_golden import import_reference, load_weights, tt  # noqa: E402

SHARE_CFG = dict(num_layers=3, share_layer_encoder=(0, 1, 0), share_layer_decoder=(0, 0, 1))


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)
    get_model, Config, losses, optim, prune = import_reference()
    cfg = Config(**dict(C.TINY_CFG, **SHARE_CFG))
    model = get_model("relation_transformer")(cfg)
    load_weights(model, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)        # named_parameters(): every shared tensor once
    model.eval()
    tb = tt(C.make_inputs(**C.G1_INPUTS))
    g = {}
    g["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
    g["param_names"] = np.array([n for n, _ in model.named_parameters()])
    g["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    model.zero_grad()
    logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
    loss.backward()
    g["logp"] = logp.detach().numpy()
    g["xe_loss"] = np.float32(loss.item())
    for n, p in model.named_parameters():
        g["grad/" + n] = p.grad.numpy().copy()
    # Cached decoding is only pinned with an UNSHARED decoder: the reference keeps the attention K/V cache on the module
    # (transformer.py:240-273, 457-469), so two positions that are the same module append to ONE cache and each sees the
    # other's keys — its incremental log-probs then differ from its own teacher-forced ones.  Encoder sharing is unaffected.
    cfg_b = Config(**dict(C.TINY_CFG, num_layers=3, share_layer_encoder=(0, 1, 0)))
    model_b = get_model("relation_transformer")(cfg_b)
    load_weights(model_b, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
    model_b.eval()
    g["enc_only/param_names"] = np.array([n for n, _ in model_b.named_parameters()])
    with torch.no_grad():
        for bs in (1, 3):
            seq_o, lp_o = model_b(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                                  opt={"beam_size": bs}, mode="sample")
            g[f"enc_only/decode_b{bs}/seq"] = seq_o.numpy()
            g[f"enc_only/decode_b{bs}/logprobs"] = lp_o.numpy()
    np.savez_compressed(os.path.join(HERE, "g8_tiny_share_layer.npz"), **g)
    print("g8: loss", float(g["xe_loss"]), "params", int(g["n_params"]), "state_dict keys", len(g["state_dict_keys"]))


if __name__ == "__main__":
    main()
