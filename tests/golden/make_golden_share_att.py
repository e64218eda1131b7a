#!/usr/bin/env python3
"""Golden fixture G9: ACORT projection sharing inside the attention modules (`share_att_encoder` / `share_att_decoder` in
{"kv", "qk"}, relation_transformer.py:140-175, transformer.py:223-263) on the tiny ORT, produced by running the REFERENCE
on CPU.  Same recipe as G1; both combinations (encoder kv + decoder qk, encoder qk + decoder kv).
    python tests/golden/make_golden_share_att.py      # writes tests/golden/g9_tiny_share_att.npz
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402
from make_golden import import_reference, load_weights, tt  # noqa: E402

CASES = {"kv_qk": dict(share_att_encoder="kv", share_att_decoder="qk"),
         "qk_kv": dict(share_att_encoder="qk", share_att_decoder="kv")}


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)
    get_model, Config, losses, optim, prune = import_reference()
    tb = tt(C.make_inputs(**C.G1_INPUTS))
    g = {}
    for tag, extra in CASES.items():
        cfg = Config(**dict(C.TINY_CFG, **extra))
        model = get_model("relation_transformer")(cfg)
        load_weights(model, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
        model.eval()
        g[tag + "/param_names"] = np.array([n for n, _ in model.named_parameters()])
        g[tag + "/n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
        model.zero_grad()
        logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
        loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
        loss.backward()
        g[tag + "/logp"] = logp.detach().numpy()
        g[tag + "/xe_loss"] = np.float32(loss.item())
        for n, p in model.named_parameters():
            g[tag + "/grad/" + n] = p.grad.numpy().copy()
        with torch.no_grad():
            for bs in (1, 3):
                seq_o, lp_o = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                                    opt={"beam_size": bs}, mode="sample")
                g[f"{tag}/decode_b{bs}/seq"] = seq_o.numpy()
                g[f"{tag}/decode_b{bs}/logprobs"] = lp_o.numpy()
        print("g9", tag, "loss", float(g[tag + "/xe_loss"]), "params", int(g[tag + "/n_params"]))
    np.savez_compressed(os.path.join(HERE, "g9_tiny_share_att.npz"), **g)


if __name__ == "__main__":
    main()
