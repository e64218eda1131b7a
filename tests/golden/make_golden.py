#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference); the GPU box only sees the .npz files.
    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference is imported with an in-memory stub for ``torchvision.transforms`` (the only missing
import on the model path). Weights and inputs come from ``common.py`` (pure numpy,
deterministic), so tests can rebuild them without the reference.
"""
import os
import sys
import types
import zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402

REF = os.environ.get("ORT_REFERENCE", "/root/reference")


def import_reference():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:  # pragma: no cover - stub
        def __init__(self, *a, **k):
            pass

    tvt.Compose = Compose
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.path.insert(0, REF)
    import torch  # noqa
    from sparse_caption.models import get_model
    from sparse_caption.utils.config import Config
    from sparse_caption.utils import losses, optim
    from sparse_caption.pruning import prune
    return get_model, Config, losses, optim, prune


def load_weights(model, seed, gen_scale, eos_bias, keep_prob=None):
    import torch
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    sd = C.state_dict_from_shapes(shapes, seed, gen_scale, eos_bias, keep_prob)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))
    return sd


def tt(batch, keys=("att_feats", "boxes", "att_masks", "seqs", "masks")):
    import torch
    return {k: torch.from_numpy(batch[k]) for k in keys}


def bits(mask_tensors):
    flat = np.concatenate([m.reshape(-1) for m in mask_tensors]).astype(np.uint8)
    return np.packbits(flat)


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)
    get_model, Config, losses, optim, prune = import_reference()
    out_dir = HERE

    # ------------------------------------------------------------------ G1: tiny dense ORT
    cfg = Config(**C.TINY_CFG)
    model = get_model("relation_transformer")(cfg)
    SEED, GEN_SCALE, EOS_BIAS = C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS
    load_weights(model, SEED, GEN_SCALE, EOS_BIAS)
    model.eval()
    batch = C.make_inputs(**C.G1_INPUTS)
    tb = tt(batch)
    g1 = {}
    with torch.no_grad():
        att, boxes, seq, amask, smask = model._prepare_feature(tb["att_feats"], tb["att_masks"], tb["boxes"], tb["seqs"])
        g1["att_embed"] = att.numpy()
        from sparse_caption.models.relation_transformer import BoxMultiHeadedAttention as BMHA
        emb = BMHA.BoxRelationalEmbedding(tb["boxes"])
        g1["box_embedding"] = emb.numpy()
        lb = []
        for layer in model.model.encoder.layers:
            sa = layer.self_attn
            g = torch.cat([l(emb.view(-1, 64)).view(emb.shape[0], 1, emb.shape[1], emb.shape[2]) for l in sa.WGs], 1)
            lb.append(torch.log(torch.clamp(torch.relu(g), min=1e-6)))
        g1["box_logbias"] = torch.stack(lb, 0).numpy()
        mem = model.model.encode(att, boxes, amask)
        g1["memory"] = mem.numpy()
    # log-probs + loss + grads (eval mode => dropout off, autograd on)
    model.zero_grad()
    logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
    loss.backward()
    g1["logp"] = logp.detach().numpy()
    g1["xe_loss"] = np.float32(loss.item())
    for n, p in model.named_parameters():
        g1["grad/" + n] = p.grad.numpy().copy()
    with torch.no_grad():
        for bs in (1, 3, 5):
            seq_o, lp_o = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                                opt={"beam_size": bs}, mode="sample")
            g1[f"decode_b{bs}/seq"] = seq_o.numpy()
            g1[f"decode_b{bs}/logprobs"] = lp_o.numpy()
            if bs > 1:
                g1[f"decode_b{bs}/p"] = np.array([[d["p"] for d in db] for db in model.done_beams], np.float32)
        # length penalty + decoding constraint variants (beam 3)
        seq_o, lp_o = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                            opt={"beam_size": 3, "length_penalty": "wu_0.7", "decoding_constraint": 1}, mode="sample")
        g1["decode_b3_wu_dc/seq"] = seq_o.numpy()
        g1["decode_b3_wu_dc/logprobs"] = lp_o.numpy()
        seq_o, lp_o = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                            opt={"beam_size": 1, "decoding_constraint": 1}, mode="sample")
        g1["decode_b1_dc/seq"] = seq_o.numpy()
        g1["decode_b1_dc/logprobs"] = lp_o.numpy()
        # multinomial rollout (torch RNG; tokens are data, the check is teacher-forced == incremental)
        torch.manual_seed(5)
        seq_s, lp_s = model(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                            opt={"num_random_sample": 2, "beam_size": 0}, mode="sample")
        g1["sample_ns2/seq"] = seq_s.numpy()
        g1["sample_ns2/logprobs"] = lp_s.numpy()
        # incremental step API
        model.apply(model.enable_incremental_decoding)
        att, boxes, _, amask, _ = model._prepare_feature(tb["att_feats"], tb["att_masks"], tb["boxes"])
        mem = model.model.encode(att, boxes, amask)
        it = torch.full((3,), C.BOS, dtype=torch.long)
        lp0, st = model.get_logprobs_state(it, mem, amask, None)
        it1 = lp0.argmax(-1)
        lp1, st = model.get_logprobs_state(it1, mem, amask, st)
        model.apply(model.disable_incremental_decoding)
        g1["step/logp0"] = lp0.numpy()
        g1["step/logp1"] = lp1.numpy()
        g1["step/state_shapes"] = np.array([list(s.shape) + [0] * (4 - s.dim()) for s in st], np.int64)
    # SCST loss on the rollout
    rs = np.random.RandomState(3)
    reward = rs.normal(size=(6,)).astype(np.float32)
    g1["scst/reward"] = reward
    g1["scst/loss"] = np.float32(losses.RewardCriterion()(lp_s, seq_s.view(-1, seq_s.size(-1)) != 0, torch.from_numpy(reward)).item())
    for k, v in batch.items():
        g1["in/" + k] = v
    g1["meta"] = np.array([SEED, 77], np.int64)
    g1["gen_scale_eos_bias"] = np.array([GEN_SCALE, EOS_BIAS], np.float32)
    np.savez_compressed(os.path.join(out_dir, "g1_tiny_dense.npz"), **g1)
    print("g1: loss", g1["xe_loss"], "greedy", g1["decode_b1/seq"][:, 0, :8].tolist())
    print("g1: beam5 lens", (g1["decode_b5/seq"] != 0).sum(-1).tolist())

    # ------------------------------------------------------------------ G4: 3 Noam/Adam/clip steps on the tiny model
    g4 = {}
    model.eval()  # dropout off, but parameters train
    cfg_opt = Config(lr_scheduler="noam", optim="adam", d_model=cfg.d_model, noamopt_factor=1.0, noamopt_warmup=10)
    opt = optim.get_optim(model.parameters(), cfg_opt)
    crit = losses.LanguageModelCriterion()
    step_losses, rates = [], []
    for step in range(3):
        opt.zero_grad()
        logp = model(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
        loss = crit(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
        loss.backward()
        optim.clip_gradient(opt, 0.1)
        opt.step(epoch=0)
        step_losses.append(loss.item())
        rates.append(opt.rate())
    g4["losses"] = np.array(step_losses, np.float32)
    g4["rates"] = np.array(rates, np.float64)
    for n, p in model.named_parameters():
        if n in ("att_embed.0.weight", "model.encoder.layers.0.self_attn.WGs.3.weight",
                 "model.encoder.layers.1.feed_forward.w_1.weight", "model.decoder.layers.1.src_attn.linears.2.bias",
                 "model.decoder.norm.a_2", "model.tgt_embed.0.lut.weight", "model.generator.proj.bias"):
            g4["param/" + n] = p.detach().numpy().copy()
    # key-projection biases have an analytically ZERO gradient (softmax shift invariance); Adam (eps 1e-9) turns their
    # rounding-noise gradients into full-size steps, so they are excluded from the checksum
    g4["param_abs_sum"] = np.float64(sum(p.detach().double().abs().sum().item() for n, p in model.named_parameters()
                                         if not n.endswith("attn.linears.1.bias")))
    np.savez_compressed(os.path.join(out_dir, "g4_tiny_optim.npz"), **g4)
    print("g4: losses", step_losses, "rates", rates)

    # ------------------------------------------------------------------ G3: tiny ORT-prune
    g3 = {}
    KEEP = C.G3_KEEP
    pcfg = Config(**C.TINY_CFG)
    pmodel = get_model("relation_transformer_prune")(pcfg)
    load_weights(pmodel, SEED, GEN_SCALE, EOS_BIAS, keep_prob=KEEP)
    pmodel.eval()
    with torch.no_grad():
        logp = pmodel(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
        g3["eval/logp"] = logp.numpy()
        # Decoding of a pruned model follows scripts/eval_model.py:64-88: COO-sparse state dict -> densify ->
        # load into the DENSE class.  (The `_prune` class itself never enables cached attention: its attention
        # modules skip CachedMultiHeadedAttention.__init__ (relation_transformer_prune.py:40-54), so they have no
        # `incremental_decoding` attribute and transformer.py:445-449 leaves them history-less.)
        from sparse_caption.utils.model_utils import densify_state_dict
        import copy
        sparse_sd = copy.deepcopy(pmodel).state_dict_sparse()
        dense_sd = densify_state_dict(sparse_sd)
        dmodel = get_model("relation_transformer")(cfg)
        dmodel.load_state_dict(dense_sd, strict=True)
        dmodel.eval()
        g3["eval/dense_roundtrip_max_abs_diff"] = np.float32((dmodel(att_feats=tb["att_feats"], boxes=tb["boxes"],
                                                       seqs=tb["seqs"], att_masks=tb["att_masks"]) - logp).abs().max().item())
        g3["eval/sparse_nnz"] = np.int64(sum(int(v._nnz()) for v in sparse_sd.values() if v.is_sparse))
        seq_o, lp_o = dmodel(att_feats=tb["att_feats"], boxes=tb["boxes"], att_masks=tb["att_masks"],
                             opt={"beam_size": 3}, mode="sample")
        g3["eval/decode_b3/seq"] = seq_o.numpy()
        g3["eval/decode_b3/logprobs"] = lp_o.numpy()
        tot, nnz, per, names = pmodel.all_mask_sparsities
        g3["sparsity/total"] = np.float64(float(tot))
        g3["sparsity/nnz"] = np.float64(float(nnz))
        g3["sparsity/per_tensor"] = np.array([float(x) for x in per], np.float64)
        g3["sparsity/names"] = np.array(list(names))
        g3["mask_avg"] = np.float64(float(pmodel.all_mask_avg))
        g3["total_weight_params"] = np.int64(pmodel.total_weight_params)
        g3["total_mask_params"] = np.int64(pmodel.total_mask_params)
        sl = []
        for step in (0, 25, 50, 100, 150):
            sl.append(float(pmodel.compute_sparsity_loss(0.9, 30.0, step, 100)))
        g3["sparsity_loss"] = np.array(sl, np.float64)
    # eval-mode gradients wrt weights and mask logits (Round straight-through + sigmoid derivative)
    pmodel.zero_grad()
    logp = pmodel(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
    loss = loss + pmodel.compute_sparsity_loss(0.9, 30.0, 50, 100)
    loss.backward()
    g3["eval/loss_total"] = np.float32(loss.item())
    for n, p in pmodel.named_parameters():
        g3["eval/grad/" + n] = (p.grad.numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32))
    pmodel.zero_grad()
    pmodel.compute_sparsity_loss(0.9, 30.0, 50, 100).backward()
    for n in ("att_embed.0.weight_pruning_mask", "model.decoder.layers.1.feed_forward.w_1.weight_pruning_mask"):
        g3["sploss_grad/" + n] = dict(pmodel.named_parameters())[n].grad.numpy().copy()

    # train mode with injected Bernoulli draws: u depends on the SHAPE of the probability tensor only
    def u_for_shape(shape):
        rs_ = np.random.RandomState(zlib.crc32(str(tuple(shape)).encode()) & 0x7FFFFFFF)
        return rs_.uniform(size=tuple(shape)).astype(np.float32)

    orig_bernoulli = torch.bernoulli
    torch.bernoulli = lambda p, *a, **k: (torch.from_numpy(u_for_shape(p.shape)) < p).to(p.dtype)
    try:
        pmodel.train()
        for m_ in pmodel.modules():
            if isinstance(m_, torch.nn.Dropout):
                m_.p = 0.0
        pmodel.zero_grad()
        logp = pmodel(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
        loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
        loss.backward()
        g3["train/logp"] = logp.detach().numpy()
        g3["train/loss"] = np.float32(loss.item())
        for n, p in pmodel.named_parameters():
            if any(s in n for s in ("att_embed.0.weight", "encoder.layers.0.self_attn.linears.1.weight",
                                    "encoder.layers.1.self_attn.WGs.2.weight", "decoder.layers.0.src_attn.linears.2.weight",
                                    "decoder.layers.1.feed_forward.w_2.weight", "tgt_embed.0.lut.weight",
                                    "generator.proj.weight")):
                g3["train/grad/" + n] = p.grad.numpy().copy()
    finally:
        torch.bernoulli = orig_bernoulli
    pmodel.eval()

    # magnitude / SNIP one-shot mask updates at sparsity 0.8
    for mtype in ("mag_blind", "mag_uniform", "mag_dist", "snip"):
        c2 = Config(**dict(C.TINY_CFG, prune_type=mtype))
        m2 = get_model("relation_transformer_prune")(c2)
        shapes = {n: tuple(p.shape) for n, p in m2.named_parameters() if not n.endswith("_pruning_mask")}
        sd = C.state_dict_from_shapes(shapes, SEED, GEN_SCALE, EOS_BIAS)
        with torch.no_grad():
            for n, p in m2.named_parameters():
                if n in sd:
                    p.copy_(torch.from_numpy(sd[n]))
        m2.eval()
        if mtype == "snip":
            m2.zero_grad()
            logp = m2(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
            loss = losses.LanguageModelCriterion()(logp, tb["seqs"][:, 1:], tb["masks"][:, 1:])
            loss.backward()
            g3["snip/grad_abs_sum"] = np.float64(sum(p.grad.double().abs().sum().item() for _, p in m2.all_pruning_masks()))
        m2.update_masks_once(0.8)
        names, masks = zip(*m2.all_pruning_masks())
        g3[f"{mtype}/mask_bits"] = bits([m.detach().numpy() for m in masks])
        g3[f"{mtype}/names"] = np.array(list(names))
        tot, nnz, per, _ = m2.all_mask_sparsities
        g3[f"{mtype}/per_tensor"] = np.array([float(x) for x in per], np.float64)
        g3[f"{mtype}/total"] = np.float64(float(tot))
        with torch.no_grad():
            logp = m2(att_feats=tb["att_feats"], boxes=tb["boxes"], seqs=tb["seqs"], att_masks=tb["att_masks"])
        g3[f"{mtype}/logp_sum_abs"] = np.float64(logp.double().abs().sum().item())
        g3[f"{mtype}/logp_row0"] = logp[0, 0].numpy()
    g3["keep_prob"] = np.float32(KEEP)
    np.savez_compressed(os.path.join(out_dir, "g3_tiny_prune.npz"), **g3)
    print("g3: sparsity", g3["sparsity/total"], "loss", g3["sparsity_loss"])

    # ------------------------------------------------------------------ G2: full-size config-1 (outputs only)
    g2 = {}
    fcfg = Config(**C.FULL_CFG)
    fmodel = get_model("relation_transformer")(fcfg)
    FSEED = C.G2_SEED
    load_weights(fmodel, FSEED, 1.0, 0.0)
    fmodel.eval()
    fb = C.make_inputs(**C.G2_INPUTS)
    ftb = tt(fb)
    fmodel.zero_grad()
    logp = fmodel(att_feats=ftb["att_feats"], boxes=ftb["boxes"], seqs=ftb["seqs"], att_masks=ftb["att_masks"])
    loss = losses.LanguageModelCriterion()(logp, ftb["seqs"][:, 1:], ftb["masks"][:, 1:])
    loss.backward()
    g2["xe_loss"] = np.float32(loss.item())
    lp = logp.detach()
    g2["logp_slice"] = lp[:, :, :32].numpy()
    g2["logp_target"] = lp.gather(2, ftb["seqs"][:, 1:].unsqueeze(2)).squeeze(2).numpy()
    g2["logp_max"] = lp.max(-1).values.numpy()
    g2["logp_argmax"] = lp.argmax(-1).numpy()
    for n, p in fmodel.named_parameters():
        g2["grad_abs_sum/" + n] = np.float64(p.grad.double().abs().sum().item())
    g2["grad/model.decoder.norm.a_2"] = fmodel.model.decoder.norm.a_2.grad.numpy().copy()
    g2["grad/model.encoder.layers.0.self_attn.WGs.0.weight"] = fmodel.model.encoder.layers[0].self_attn.WGs[0].weight.grad.numpy().copy()
    g2["grad/att_embed.0.bias"] = fmodel.att_embed[0].bias.grad.numpy().copy()
    with torch.no_grad():
        for bs in (1, 5):
            seq_o, lp_o = fmodel(att_feats=ftb["att_feats"], boxes=ftb["boxes"], att_masks=ftb["att_masks"],
                                 opt={"beam_size": bs}, mode="sample")
            g2[f"decode_b{bs}/seq"] = seq_o.numpy()
            g2[f"decode_b{bs}/logprobs"] = lp_o.numpy()
    g2["meta"] = np.array([FSEED, 99], np.int64)
    np.savez_compressed(os.path.join(out_dir, "g2_full_cfg1.npz"), **g2)
    print("g2: loss", g2["xe_loss"], "greedy", g2["decode_b1/seq"][0, 0, :6].tolist())
    for f in sorted(os.listdir(out_dir)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(out_dir, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
