"""Pins the oracle (oracle/ort_oracle.py) to golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only; this is the `parity pinned` gate of the oracle."""
import zlib

import numpy as np
import pytest
import torch

import common as C
import helpers as H
from oracle import ort_oracle as O

TOL = 2e-5


def _cfg(d):
    return O.OCfg(**{k: v for k, v in d.items() if not k.startswith("prune")})


def close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


@pytest.fixture(scope="module")
def g1(golden):
    return golden("g1_tiny_dense")


def test_inputs_rebuild(g1):
    b = C.make_inputs(**C.G1_INPUTS)
    for k, v in b.items():
        np.testing.assert_array_equal(v, g1["in/" + k])


def test_feature_prep_and_geometry(g1):
    P, b, cfg = H.g1_state(), H.g1_batch(), _cfg(C.TINY_CFG)
    close(O.att_embed(P, b["att_feats"], b["att_masks"]), g1["att_embed"])
    emb = O.box_relational_embedding(b["boxes"])
    close(emb, g1["box_embedding"], 1e-6)
    lb = torch.stack([O.box_logbias(P, l, emb, cfg.num_heads) for l in range(cfg.num_layers)], 0)
    # compare g = exp(logbias): log() of a near-zero relu output amplifies 1-ulp dot-product differences
    close(lb.exp(), np.exp(g1["box_logbias"]), 2e-6)
    close(lb, g1["box_logbias"], 2e-3)
    close(O.encode(P, cfg, b["att_feats"], b["boxes"], b["att_masks"]), g1["memory"])


def test_teacher_forcing_loss_and_grads(g1):
    P, b, cfg = H.g1_state(requires_grad=True), H.g1_batch(), _cfg(C.TINY_CFG)
    logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g1["logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g1["xe_loss"])) < 1e-5
    loss.backward()
    for n, p in P.items():
        close(p.grad, g1["grad/" + n], 5e-5)


@pytest.mark.parametrize("bs", [1, 3, 5])
def test_decode(g1, bs):
    P, b, cfg = H.g1_state(), H.g1_batch(), _cfg(C.TINY_CFG)
    with torch.no_grad():
        if bs == 1:
            seq, lp = O.sample_greedy_or_multinomial(P, cfg, b["att_feats"], b["boxes"], b["att_masks"])
        else:
            seq, lp, p = O.beam_search(P, cfg, b["att_feats"], b["boxes"], b["att_masks"], bs)
            close(p, g1[f"decode_b{bs}/p"], 1e-4)
    np.testing.assert_array_equal(seq.numpy(), g1[f"decode_b{bs}/seq"])
    close(lp, g1[f"decode_b{bs}/logprobs"], 1e-4)


def test_decode_options(g1):
    P, b, cfg = H.g1_state(), H.g1_batch(), _cfg(C.TINY_CFG)
    with torch.no_grad():
        seq, lp, _ = O.beam_search(P, cfg, b["att_feats"], b["boxes"], b["att_masks"], 3,
                                   decoding_constraint=1, length_penalty="wu_0.7")
        np.testing.assert_array_equal(seq.numpy(), g1["decode_b3_wu_dc/seq"])
        close(lp, g1["decode_b3_wu_dc/logprobs"], 1e-4)
        seq, lp = O.sample_greedy_or_multinomial(P, cfg, b["att_feats"], b["boxes"], b["att_masks"],
                                                 decoding_constraint=1)
        np.testing.assert_array_equal(seq.numpy(), g1["decode_b1_dc/seq"])
        close(lp, g1["decode_b1_dc/logprobs"], 1e-4)


def test_step_api_and_incremental_equals_teacher_forced(g1):
    P, b, cfg = H.g1_state(), H.g1_batch(), _cfg(C.TINY_CFG)
    with torch.no_grad():
        mem = O.encode(P, cfg, b["att_feats"], b["boxes"], b["att_masks"])
        st = O.DecodeState(P, cfg, mem, b["att_masks"])
        lp0 = O.decode_step(st, torch.full((3,), C.BOS, dtype=torch.long))
        lp1 = O.decode_step(st, lp0.argmax(-1))
        close(lp0, g1["step/logp0"], 5e-5)
        close(lp1, g1["step/logp1"], 5e-5)
        # reference state = [ys] + 24 caches shaped (h, rows, len, d_k): 2 layers -> 1 + 8 entries
        assert g1["step/state_shapes"].shape[0] == 1 + 4 * cfg.num_layers
        # G5: log-probs of a sampled rollout recomputed by ONE teacher-forced pass
        seq = torch.from_numpy(g1["sample_ns2/seq"])  # (3,2,18)
        rows = seq.view(-1, 18)
        tf_in = torch.cat([torch.full((rows.size(0), 1), C.BOS, dtype=torch.long), rows], 1)  # (6,19)
        logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], tf_in, b["att_masks"])  # (6,18,V)
        tok_lp = logp.gather(2, rows.unsqueeze(2)).squeeze(2)
        ref = torch.from_numpy(g1["sample_ns2/logprobs"]).view(-1, 18)
        valid = rows != 0
        assert (tok_lp - ref)[valid].abs().max() < 2e-5
        loss = O.reward_loss(torch.where(valid, tok_lp, ref), seq, torch.from_numpy(g1["scst/reward"]))
        assert abs(loss.item() - float(g1["scst/loss"])) < 1e-5


def test_gumbel_sampler_is_a_sampler():
    g = O.gumbel_from_hash(seed=3, t=0, rows=4000, vocab=4)
    logp = torch.log(torch.tensor([0.1, 0.2, 0.3, 0.4]))
    freq = torch.bincount((logp + g).argmax(-1), minlength=4).float() / 4000
    assert (freq - logp.exp()).abs().max() < 0.03


def test_noam_adam_clip(golden):
    g4 = golden("g4_tiny_optim")
    P, b, cfg = H.g1_state(requires_grad=True), H.g1_batch(), _cfg(C.TINY_CFG)
    state = {}
    for step in range(3):
        for p in P.values():
            p.grad = None
        logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
        loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
        loss.backward()
        assert abs(loss.item() - float(g4["losses"][step])) < 2e-4
        lr = O.noam_rate(step + 1, cfg.d_model, 1.0, 10)
        assert abs(lr - float(g4["rates"][step])) < 1e-12
        with torch.no_grad():
            O.adam_clip_step({n: p for n, p in P.items()}, {n: p.grad for n, p in P.items()}, state, lr, clip=0.1)
    for k, v in g4.items():
        if k.startswith("param/"):
            close(P[k[6:]], v, 2e-4)
    tot = sum(p.detach().double().abs().sum().item() for n, p in P.items() if not n.endswith("attn.linears.1.bias"))
    assert abs(tot - float(g4["param_abs_sum"])) / tot < 1e-5


# ------------------------------------------------------------------------------------------ prune
@pytest.fixture(scope="module")
def g3(golden):
    return golden("g3_tiny_prune")


def _prune_state(requires_grad=False):
    return H.torch_state(H.prune_param_shapes(C.TINY_CFG), C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS,
                         keep_prob=C.G3_KEEP, requires_grad=requires_grad)


def test_prune_eval_forward_stats_and_grads(g3):
    P, b, cfg = _prune_state(requires_grad=True), H.g1_batch(), _cfg(C.TINY_CFG)
    E = O.effective_params(P, "supermask", training=False)
    logp = O.forward_logp(E, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g3["eval/logp"], 5e-5)
    with torch.no_grad():
        seq, lp, _ = O.beam_search(E, cfg, b["att_feats"], b["boxes"], b["att_masks"], 3)
    np.testing.assert_array_equal(seq.numpy(), g3["eval/decode_b3/seq"])
    tot, nnz, per, names = O.mask_sparsities(P, "supermask")
    assert abs(float(tot) - float(g3["sparsity/total"])) < 1e-6
    assert float(nnz) == float(g3["sparsity/nnz"])
    assert sorted(names) == sorted(g3["sparsity/names"].tolist())
    ref_per = dict(zip(g3["sparsity/names"].tolist(), g3["sparsity/per_tensor"]))
    for n, s in zip(names, per):
        assert abs(float(s) - ref_per[n]) < 1e-6
    for i, step in enumerate((0, 25, 50, 100, 150)):
        assert abs(float(O.sparsity_loss(P, 0.9, 30.0, step, 100)) - g3["sparsity_loss"][i]) < 1e-5
    assert sum(v.numel() for k, v in P.items() if O.is_mask(k)) == int(g3["total_mask_params"])
    assert sum(v.numel() for k, v in P.items() if not O.is_mask(k)) == int(g3["total_weight_params"])
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:]) + O.sparsity_loss(P, 0.9, 30.0, 50, 100)
    assert abs(loss.item() - float(g3["eval/loss_total"])) < 1e-4
    loss.backward()
    for n, p in P.items():
        close(p.grad if p.grad is not None else torch.zeros_like(p), g3["eval/grad/" + n], 5e-5)
    # the sparsity loss alone: straight-through Round over sigmoid (tiny values -> relative check)
    for p in P.values():
        p.grad = None
    O.sparsity_loss(P, 0.9, 30.0, 50, 100).backward()
    for k, v in g3.items():
        if k.startswith("sploss_grad/"):
            np.testing.assert_allclose(P[k[len("sploss_grad/"):]].grad.numpy(), v, rtol=1e-4, atol=1e-12)


def test_prune_train_injected_bernoulli(g3):
    P, b, cfg = _prune_state(requires_grad=True), H.g1_batch(), _cfg(C.TINY_CFG)
    samples = {}
    for n, p in P.items():
        if O.is_mask(n):
            u = np.random.RandomState(zlib.crc32(str(tuple(p.shape)).encode()) & 0x7FFFFFFF).uniform(size=tuple(p.shape))
            samples[n] = (torch.from_numpy(u.astype(np.float32)) < torch.sigmoid(p.detach())).float()
    E = O.effective_params(P, "supermask", training=True, samples=samples)
    logp = O.forward_logp(E, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g3["train/logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g3["train/loss"])) < 1e-5
    loss.backward()
    n_checked = 0
    for k, v in g3.items():
        if k.startswith("train/grad/"):
            close(P[k[len("train/grad/"):]].grad, v, 5e-5)
            n_checked += 1
    assert n_checked >= 10


@pytest.mark.parametrize("mtype", ["mag_blind", "mag_uniform", "mag_dist", "snip"])
def test_update_masks_once(g3, mtype):
    shapes = H.prune_param_shapes(C.TINY_CFG)
    P = H.torch_state({k: v for k, v in shapes.items() if not O.is_mask(k)}, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
    for k, shp in shapes.items():
        if O.is_mask(k):
            P[k] = torch.ones(shp)
    b, cfg = H.g1_batch(), _cfg(C.TINY_CFG)
    grads = None
    if mtype == "snip":
        for k in P:
            if O.is_mask(k):
                P[k].requires_grad_(True)
        E = O.effective_params(P, "snip")
        logp = O.forward_logp(E, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
        O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:]).backward()
        grads = {k: P[k].grad for k in P if O.is_mask(k)}
        tot = sum(g.double().abs().sum().item() for g in grads.values())
        assert abs(tot - float(g3["snip/grad_abs_sum"])) / tot < 1e-4
        P = {k: v.detach() for k, v in P.items()}
    new = O.update_masks_once(P, mtype, 0.8, grads)
    names = g3[f"{mtype}/names"].tolist()
    ref = H.unpack_bits(g3[f"{mtype}/mask_bits"], [shapes[n] for n in names])
    mism = sum(int((new[n].numpy() != r).sum()) for n, r in zip(names, ref))
    total = sum(r.size for r in ref)
    # SNIP saliencies are sums of fp32 products: allow a handful of threshold-adjacent flips
    assert mism <= (8 if mtype == "snip" else 0), f"{mism}/{total} mask bits differ"
    P.update(new)
    tot, _, per, _ = O.mask_sparsities(P, mtype, names)
    assert abs(float(tot) - float(g3[f"{mtype}/total"])) < 1e-4
    if mtype != "snip":
        E = O.effective_params(P, mtype)
        with torch.no_grad():
            logp = O.forward_logp(E, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
        close(logp[0, 0], g3[f"{mtype}/logp_row0"], 5e-5)


def test_gradual_schedule():
    assert O.gradual_sparsity(0.95, 999, 1000, 10) is None
    assert O.gradual_sparsity(0.95, 1000, 1000, 10) == pytest.approx(0.0)
    assert O.gradual_sparsity(0.95, 6000, 1000, 10) == pytest.approx(0.95 - 0.95 * 0.5 ** 3)
    assert O.gradual_sparsity(0.95, 11000, 1000, 10) == pytest.approx(0.95)
    assert O.gradual_sparsity(0.95, 12000, 1000, 10) is None


# ------------------------------------------------------------------------------------------ full size (config 1)
def test_full_size_config1(golden):
    g2 = golden("g2_full_cfg1")
    P = H.torch_state(H.dense_param_shapes(C.FULL_CFG), C.G2_SEED, requires_grad=True)
    b, cfg = H.torch_batch(C.make_inputs(**C.G2_INPUTS)), _cfg(C.FULL_CFG)
    torch.set_num_threads(8)
    logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g2["xe_loss"])) < 1e-5
    close(logp[:, :, :32], g2["logp_slice"], 1e-4)
    close(logp.gather(2, b["seqs"][:, 1:].unsqueeze(2)).squeeze(2), g2["logp_target"], 1e-4)
    np.testing.assert_array_equal(logp.argmax(-1).numpy(), g2["logp_argmax"])
    loss.backward()
    for n, p in P.items():
        ref = float(g2["grad_abs_sum/" + n])
        assert abs(p.grad.double().abs().sum().item() - ref) <= 2e-4 * max(ref, 1e-3), n
    close(P["model.decoder.norm.a_2"].grad, g2["grad/model.decoder.norm.a_2"], 1e-4)
    close(P["att_embed.0.bias"].grad, g2["grad/att_embed.0.bias"], 1e-4)
    Pd = {k: v.detach() for k, v in P.items()}
    with torch.no_grad():
        seq, lp = O.sample_greedy_or_multinomial(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"])
        np.testing.assert_array_equal(seq.numpy(), g2["decode_b1/seq"])
        seq, lp, _ = O.beam_search(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"], 5)
        np.testing.assert_array_equal(seq.numpy(), g2["decode_b5/seq"])
        close(lp, g2["decode_b5/logprobs"], 2e-4)


def test_non_trigonometric_box_embedding_vs_reference_golden(golden):
    """G5: `no_box_trigonometric_embedding` (4-d geometry embedding, relation_transformer.py:131-136,243-256)."""
    g5 = golden("g5_tiny_notrig")
    cfgd = dict(C.TINY_CFG, no_box_trigonometric_embedding=True)
    cfg = _cfg(cfgd)
    P = H.torch_state(H.dense_param_shapes(cfgd), C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, requires_grad=True)
    b = H.g1_batch()
    logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g5["logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g5["xe_loss"])) < 1e-5
    loss.backward()
    for k in list(g5.keys()):
        if k.startswith("grad/"):
            np.testing.assert_allclose(P[k[5:]].grad.numpy(), g5[k], rtol=2e-3, atol=2e-5, err_msg=k)
    with torch.no_grad():
        Pd = {k: v.detach() for k, v in P.items()}
        seq, lp, _ = O.beam_search(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"], beam_size=3)
        np.testing.assert_array_equal(seq.numpy(), g5["decode_b3/seq"])


def test_share_layer_vs_reference_golden(golden):
    """G8: ACORT layer sharing — positions that share a module read the same tensors; their gradients add up."""
    g8 = golden("g8_tiny_share_layer")
    cfgd = dict(C.TINY_CFG, num_layers=3, share_layer_encoder=(0, 1, 0), share_layer_decoder=(0, 0, 1))
    cfg = _cfg({k: v for k, v in cfgd.items() if not k.startswith("share_")})
    names = [str(n) for n in g8["param_names"]]
    P = H.shared_layer_state(cfgd, names, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, requires_grad=True)
    b = H.g1_batch()
    logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g8["logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g8["xe_loss"])) < 1e-5
    loss.backward()
    for n in names:
        ref = g8["grad/" + n]
        np.testing.assert_allclose(P[n].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=n)
    # cached decoding is pinned with encoder sharing only (the reference's shared decoder modules share ONE K/V cache between
    # their positions: make_golden_share.py)
    cfgb = dict(C.TINY_CFG, num_layers=3, share_layer_encoder=(0, 1, 0))
    Pb = H.shared_layer_state(cfgb, [str(n) for n in g8["enc_only/param_names"]], C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
    with torch.no_grad():
        seq, lp, _ = O.beam_search(Pb, cfg, b["att_feats"], b["boxes"], b["att_masks"], beam_size=3)
        np.testing.assert_array_equal(seq.numpy(), g8["enc_only/decode_b3/seq"])
        close(lp, g8["enc_only/decode_b3/logprobs"], 1e-4)


@pytest.mark.parametrize("tag,enc,dec", [("kv_qk", "kv", "qk"), ("qk_kv", "qk", "kv")])
def test_share_att_vs_reference_golden(golden, tag, enc, dec):
    """G9: ACORT projection sharing (K = V from one linear, or Q and K from one linear), three linears per module."""
    g9 = golden("g9_tiny_share_att")
    cfgd = dict(C.TINY_CFG, share_att_encoder=enc, share_att_decoder=dec)
    cfg = _cfg(cfgd)
    names = [str(n) for n in g9[tag + "/param_names"]]
    shapes = H.dense_param_shapes(cfgd)
    assert set(names) == set(shapes) and sum(int(np.prod(v)) for v in shapes.values()) == int(g9[tag + "/n_params"])
    P = H.torch_state(shapes, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, requires_grad=True)
    b = H.g1_batch()
    logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
    close(logp, g9[tag + "/logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g9[tag + "/xe_loss"])) < 1e-5
    loss.backward()
    for n in names:
        ref = g9[f"{tag}/grad/{n}"]
        np.testing.assert_allclose(P[n].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=n)
    with torch.no_grad():
        Pd = {k: v.detach() for k, v in P.items()}
        for bs in (1, 3):
            if bs == 1:
                seq, lp = O.sample_greedy_or_multinomial(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"])[:2]
            else:
                seq, lp, _ = O.beam_search(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"], beam_size=3)
            np.testing.assert_array_equal(seq.numpy(), g9[f"{tag}/decode_b{bs}/seq"])
            close(lp, g9[f"{tag}/decode_b{bs}/logprobs"], 1e-4)


def test_plain_transformer_vs_reference_golden(golden):
    """G10: the plain `transformer` class — no geometry bias, padded regions embedded, `core.*` names."""
    g10 = golden("g10_tiny_plain_transformer")
    cfg = _cfg(dict(C.TINY_CFG, plain=True))
    shapes = H.plain_shapes(g10)
    assert int(g10["n_params"]) == sum(int(np.prod(v)) for v in shapes.values()) and not any(".WGs." in n for n in shapes)
    state = H.torch_state(shapes, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, requires_grad=True)
    P = O.plain_state(state)
    b = H.g1_batch()
    with torch.no_grad():
        close(O.encode({k: v.detach() for k, v in P.items()}, cfg, b["att_feats"], None, b["att_masks"]), g10["memory"], 2e-5)
    logp = O.forward_logp(P, cfg, b["att_feats"], None, b["seqs"], b["att_masks"])
    close(logp, g10["logp"], 5e-5)
    loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
    assert abs(loss.item() - float(g10["xe_loss"])) < 1e-5
    loss.backward()
    for n in shapes:
        ref = g10["grad/" + n]
        np.testing.assert_allclose(state[n].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=n)
    with torch.no_grad():
        Pd = {k: v.detach() for k, v in P.items()}
        seq, lp = O.sample_greedy_or_multinomial(Pd, cfg, b["att_feats"], None, b["att_masks"])[:2]
        np.testing.assert_array_equal(seq.numpy(), g10["decode_b1/seq"])
        seq, lp, _ = O.beam_search(Pd, cfg, b["att_feats"], None, b["att_masks"], beam_size=3)
        np.testing.assert_array_equal(seq.numpy(), g10["decode_b3/seq"])
        close(lp, g10["decode_b3/logprobs"], 1e-4)
