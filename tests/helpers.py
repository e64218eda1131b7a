"""Test-side helpers: rebuild the deterministic weights / inputs of the golden fixtures (no reference needed)."""
import numpy as np
import torch

import common as C  # tests/golden/common.py (on sys.path via conftest)


def dense_param_shapes(cfg: dict) -> dict:
    """state_dict parameter names -> shapes of the dense ORT, in reference registration order."""
    d, ff, L, h = cfg["d_model"], cfg["dim_feedforward"], cfg["num_layers"], cfg["num_heads"]
    V, Fs = cfg["vocab_size"], cfg["att_feat_size"]
    s = {"att_embed.0.weight": (d, Fs), "att_embed.0.bias": (d,)}

    def attn(pre, box):
        share = cfg.get("share_att_encoder" if box else "share_att_decoder")
        for i in range(3 if share else 4):          # relation_transformer.py:142, transformer.py:225
            s[f"{pre}.linears.{i}.weight"] = (d, d)
            s[f"{pre}.linears.{i}.bias"] = (d,)
        if box:
            for i in range(h):
                s[f"{pre}.WGs.{i}.weight"] = (1, 4 if cfg.get("no_box_trigonometric_embedding") else 64)
                s[f"{pre}.WGs.{i}.bias"] = (1,)

    def ffn(pre):
        s[pre + ".w_1.weight"] = (ff, d); s[pre + ".w_1.bias"] = (ff,)
        s[pre + ".w_2.weight"] = (d, ff); s[pre + ".w_2.bias"] = (d,)

    def norm(pre):
        s[pre + ".a_2"] = (d,); s[pre + ".b_2"] = (d,)

    for l in range(L):
        p = f"model.encoder.layers.{l}"
        attn(p + ".self_attn", True); ffn(p + ".feed_forward")
        norm(p + ".sublayer.0.norm"); norm(p + ".sublayer.1.norm")
    norm("model.encoder.norm")
    for l in range(L):
        p = f"model.decoder.layers.{l}"
        attn(p + ".self_attn", False); attn(p + ".src_attn", False); ffn(p + ".feed_forward")
        for j in range(3):
            norm(f"{p}.sublayer.{j}.norm")
    norm("model.decoder.norm")
    s["model.tgt_embed.0.lut.weight"] = (V, d)
    s["model.generator.proj.weight"] = (V, d); s["model.generator.proj.bias"] = (V,)
    return s


def prune_param_shapes(cfg: dict) -> dict:
    s = {}
    for n, shp in dense_param_shapes(cfg).items():
        s[n] = shp
        if len(shp) >= 2:
            s[n + "_pruning_mask"] = shp
    return s


def np_state(shapes, seed, gen_scale=1.0, eos_bias=0.0, keep_prob=None):
    return C.state_dict_from_shapes(shapes, seed, gen_scale, eos_bias, keep_prob)


def torch_state(shapes, seed, gen_scale=1.0, eos_bias=0.0, keep_prob=None, requires_grad=False):
    sd = np_state(shapes, seed, gen_scale, eos_bias, keep_prob)
    return {k: torch.from_numpy(v).requires_grad_(requires_grad) for k, v in sd.items()}


def torch_batch(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def g1_state(requires_grad=False):
    return torch_state(dense_param_shapes(C.TINY_CFG), C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS, requires_grad=requires_grad)


def g1_batch():
    return torch_batch(C.make_inputs(**C.G1_INPUTS))


def unpack_bits(bits, shapes):
    flat = np.unpackbits(bits)
    out, off = [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        out.append(flat[off:off + n].reshape(shp).astype(np.float32))
        off += n
    return out


def shared_layer_state(cfg, param_names, seed, gen_scale=1.0, eos_bias=0.0, requires_grad=False):
    """Weights of a `share_layer_*` model: tensors for the reference's named_parameters() (each shared tensor once, under
    its first position's name) + alias entries so that EVERY layer position has its keys (the oracle indexes by position)."""
    shapes = dense_param_shapes(cfg)
    state = torch_state({n: shapes[n] for n in param_names}, seed, gen_scale, eos_bias, requires_grad=requires_grad)
    L = cfg["num_layers"]
    for stack, key in (("encoder", "share_layer_encoder"), ("decoder", "share_layer_decoder")):
        ids = list(cfg.get(key) or range(L))
        first = {}
        for l, v in enumerate(ids):
            first.setdefault(v, l)
        for l, v in enumerate(ids):
            if first[v] != l:
                src, dst = f"model.{stack}.layers.{first[v]}.", f"model.{stack}.layers.{l}."
                for n in list(state):
                    if n.startswith(src):
                        state[dst + n[len(src):]] = state[n]
    return state


def plain_shapes(g10) -> dict:
    """Parameter name -> shape of the reference's plain `transformer` model, as recorded in golden G10."""
    return {str(n): tuple(int(x) for x in str(sh).split(",")) for n, sh in zip(g10["param_names"], g10["param_shapes"])}


FP32_EPS = 2.0 ** -24


def atomics_bar(n_addends, scale):
    """How far two runs of ONE fp32 sum of `n_addends` terms may lie apart when only the ORDER of its atomic adds differs
    (split-K weight gradients, bias / LayerNorm-parameter column sums, embedding rows): every add rounds by at most eps times a
    partial sum, and a partial sum of terms bounded by `scale / n` each stays below `scale`, so each run is within n * eps * scale
    of the exact sum and two runs within twice that.  `scale`: the magnitude of the largest gradient element; `n_addends`: an upper
    bound on the atomic contributions to one element (rows of the reduction).  A bar DERIVED from the addend count, not tuned to a
    box: a wrong mask, a lost tile or a stale operand moves gradients by O(scale), orders of magnitude above it."""
    return 2.0 * float(n_addends) * FP32_EPS * float(scale)
