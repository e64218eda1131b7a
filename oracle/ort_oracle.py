"""ORACLE — CPU restatement of the reference's Object Relation Transformer hot path.

*** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the shipped package (``sparse-image-captioning_amd``) never does and fails loudly when
its HIP library is missing.

What it is: a functional (no nn.Module) fp32 restatement, written with plain ``torch`` CPU tensor
ops, of the arithmetic of jiahuei/sparse-image-captioning's ORT path.  The reference itself is 100 %
PyTorch/ATen, so torch-on-CPU *is* the reference's kernel layer (SURVEY.md §8c "Third-party
arithmetic"); autograd supplies the backward pass the reference gets the same way.

Parity pinning: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against the
fixtures in ``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` produced by importing and
running the real reference (/root/reference) in the build container.

Parameter naming follows the reference ``state_dict`` keys, e.g.
``model.encoder.layers.0.self_attn.linears.2.weight``.  ``P`` below is a ``dict[str, Tensor]``.

Reference citations are relative to /root/reference/.
"""
import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NEG = -1e9  # masking constant, models/transformer.py:291, models/relation_transformer.py:278


# --------------------------------------------------------------------------------------------- config
class OCfg:
    """Plain attribute bag with the fields models/transformer.py:418-437 reads."""

    def __init__(self, **kw):
        self.d_model = 512
        self.dim_feedforward = 2048
        self.num_layers = 6
        self.num_heads = 8
        self.max_seq_length = 18
        self.att_feat_size = 2048
        self.vocab_size = 10001
        self.bos_token_id, self.eos_token_id, self.unk_token_id, self.pad_token_id = 2, 3, 1, 0
        self.no_box_trigonometric_embedding = False
        self.share_att_encoder = self.share_att_decoder = None      # None | "kv" | "qk" (ACORT)
        # True: the plain `transformer` model (transformer.py:617-665) — no geometry bias, padded regions embedded like the
        # others.  ``P`` then still uses this file's names (`att_embed.0.*` for `core.src_embed.0.*`, `model.` for `core.`:
        # see `plain_state`).
        self.plain = False
        for k, v in kw.items():
            setattr(self, k, v)


# --------------------------------------------------------------------------------------------- blocks
def layer_norm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """models/transformer.py:338-341 — Bessel-corrected std, eps added to the STD (not the variance)."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)  # unbiased
    return a * (x - mean) / (std + eps) + b


def box_relational_embedding(boxes: Tensor, trig: bool = True) -> Tensor:
    """models/relation_transformer.py:196-256.  boxes (B,S,4) relative [x0,y0,x1,y1] -> (B,S,S,64|4).

    The fp32 op order (``100*p`` first, then ``* 1/1000^(k/8)``) is kept: arguments reach ~690 rad,
    where one ulp of the argument already moves sin/cos by 6e-5.
    """
    x0, y0, x1, y1 = boxes.unbind(-1)
    cx = (x0 + x1) * 0.5
    cy = (y0 + y1) * 0.5
    w = (x1 - x0) + 1.0
    h = (y1 - y0) + 1.0
    # entry [b,i,j]: i = "this" box (divides by its own w/h), j = other box
    dx = torch.log(torch.clamp(torch.abs((cx[:, :, None] - cx[:, None, :]) / w[:, :, None]), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy[:, :, None] - cy[:, None, :]) / h[:, :, None]), min=1e-3))
    dw = torch.log(w[:, :, None] / w[:, None, :])
    dh = torch.log(h[:, :, None] / h[:, None, :])
    pos = torch.stack((dx, dy, dw, dh), -1)  # (B,S,S,4)
    if not trig:
        return pos
    k = torch.arange(8, dtype=boxes.dtype)
    dim_mat = 1.0 / torch.pow(torch.tensor(1000.0, dtype=boxes.dtype), k / 8.0)  # (8,)
    mul = (100.0 * pos)[..., None] * dim_mat  # (B,S,S,4,8)
    mul = mul.reshape(*pos.shape[:3], 32)
    return torch.cat((torch.sin(mul), torch.cos(mul)), -1)


def box_logbias(P: Dict[str, Tensor], layer: int, emb: Tensor, n_heads: int) -> Tensor:
    """log(clamp(relu(WG_h . e + b_h), 1e-6)) -> (B,h,S,S); relation_transformer.py:177-183,286."""
    pre = f"model.encoder.layers.{layer}.self_attn.WGs."
    W = torch.cat([P[f"{pre}{h}.weight"] for h in range(n_heads)], 0)  # (h, 64)
    b = torch.cat([P[f"{pre}{h}.bias"] for h in range(n_heads)], 0)  # (h,)
    g = torch.relu(torch.einsum("bijk,hk->bhij", emb, W) + b[None, :, None, None])
    return torch.log(torch.clamp(g, min=1e-6))


def _heads(x: Tensor, h: int) -> Tensor:
    n, l, d = x.shape
    return x.view(n, l, h, d // h).transpose(1, 2)


def _linear(P, prefix, x):
    return F.linear(x, P[prefix + ".weight"], P[prefix + ".bias"])


def project_qkv(P, prefix: str, share_att, xq: Tensor, xkv: Tensor, h: int, need_kv: bool = True):
    """Q / K / V heads and the name of the output projection of one attention module.
    share_att None: linears.0/1/2, output linears.3.  "kv": K = V = linears.1(key).  "qk": K = linears.0(key),
    V = linears.1(value); output linears.2 in both (relation_transformer.py:142,162-175; transformer.py:225,258-263)."""
    assert share_att in (None, "kv", "qk"), f"Invalid `share_att`: {share_att}"
    q = _heads(_linear(P, prefix + "linears.0", xq), h)
    out = prefix + ("linears.2" if share_att else "linears.3")
    if not need_kv:
        return q, None, None, out
    if share_att == "kv":
        k = _heads(_linear(P, prefix + "linears.1", xkv), h)
        v = k
    elif share_att == "qk":
        k = _heads(_linear(P, prefix + "linears.0", xkv), h)
        v = _heads(_linear(P, prefix + "linears.1", xkv), h)
    else:
        k = _heads(_linear(P, prefix + "linears.1", xkv), h)
        v = _heads(_linear(P, prefix + "linears.2", xkv), h)
    return q, k, v, out


def _drop(drop, site: str, x: Tensor) -> Tensor:
    """Training-mode dropout at a named site.  `drop` is None (eval: identity) or a callable (site, x) -> x * keep / (1 - p)
    with an explicit keep mask: torch's RNG stream cannot be matched, so train-mode parity INJECTS the masks (the HIP path's
    counter hash, read back through ortk_dropout_apply in tests/test_gpu_model.py::test_train_mode_dropout_vs_oracle)."""
    return x if drop is None else drop(site, x)


def attention(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], bias: Optional[Tensor] = None, drop=None, site: str = "") -> Tensor:
    """transformer.py:285-295 / relation_transformer.py:258-293 (mask fill BEFORE the additive log-bias; dropout on the
    probabilities, transformer.py:293-294)."""
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1))
    if mask is not None:
        scores = scores.masked_fill(mask == 0, NEG)
    if bias is not None:
        scores = bias + scores
    p = _drop(drop, site, torch.softmax(scores, dim=-1))
    return torch.matmul(p, v)


def _merge(x: Tensor) -> Tensor:
    n, h, l, dk = x.shape
    return x.transpose(1, 2).reshape(n, l, h * dk)


def att_embed(P, att_feats: Tensor, att_masks: Tensor, drop=None) -> Tensor:
    """Dropout(relu(Linear(att_feats))) on valid regions (relation_transformer.py:331-333), exact zeros on padded ones
    (model_utils.py:149-168)."""
    y = _drop(drop, "src", torch.relu(_linear(P, "att_embed.0", att_feats)))
    return y * (att_masks != 0).to(y.dtype)[..., None]


def encode(P, cfg, att_feats: Tensor, boxes: Tensor, att_masks: Tensor, drop=None) -> Tensor:
    """relation_transformer.py:341-365 (feature prep) + :92-113,148-191 (encoder). Returns memory (B,S,d).
    Dropout sites (drop != None): SublayerConnection x + dropout(sublayer(norm(x))) (transformer.py:356-358), the FFN's
    w_2(dropout(relu(w_1 x))) (transformer.py:324-325), the attention probabilities."""
    h = cfg.num_heads
    plain = getattr(cfg, "plain", False)
    if plain:   # transformer.py:627-629: Linear + ReLU (+ Dropout) on every row; boxes unused
        x = _drop(drop, "src", torch.relu(_linear(P, "att_embed.0", att_feats)))
        emb = None
    else:
        x = att_embed(P, att_feats, att_masks, drop)
        emb = box_relational_embedding(boxes, not cfg.no_box_trigonometric_embedding)
    kmask = (att_masks != 0)[:, None, None, :]  # (B,1,1,S)
    for l in range(cfg.num_layers):
        pre = f"model.encoder.layers.{l}."
        y = layer_norm(x, P[pre + "sublayer.0.norm.a_2"], P[pre + "sublayer.0.norm.b_2"])
        q, k, v, out = project_qkv(P, pre + "self_attn.", getattr(cfg, "share_att_encoder", None), y, y, h)
        o = attention(q, k, v, kmask, None if plain else box_logbias(P, l, emb, h), drop, f"enc{l}.att")
        x = x + _drop(drop, f"enc{l}.sub0", _linear(P, out, _merge(o)))
        y = layer_norm(x, P[pre + "sublayer.1.norm.a_2"], P[pre + "sublayer.1.norm.b_2"])
        hid = _drop(drop, f"enc{l}.ffn", torch.relu(_linear(P, pre + "feed_forward.w_1", y)))
        x = x + _drop(drop, f"enc{l}.sub1", _linear(P, pre + "feed_forward.w_2", hid))
    return layer_norm(x, P["model.encoder.norm.a_2"], P["model.encoder.norm.b_2"])


def plain_state(state: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """state_dict of the plain `transformer` class (`core.*`) under the names this file indexes."""
    out = {}
    for k, v in state.items():
        k = k.replace("core.src_embed.0.", "att_embed.0.", 1) if k.startswith("core.src_embed.0.") else k
        out["model." + k[len("core."):] if k.startswith("core.") else k] = v
    return out


def positional_encoding(n_pos: int, d: int) -> Tensor:
    """transformer.py:369-374 (fp32 construction)."""
    pe = torch.zeros(n_pos, d)
    position = torch.arange(0, n_pos).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def embed_tokens(P, cfg, tok: Tensor, pos0: int = 0) -> Tensor:
    """transformer.py:383-401: lut[tok]*sqrt(d) + pe[pos]."""
    d = cfg.d_model
    pe = positional_encoding(cfg.max_seq_length + 2, d)
    e = F.embedding(tok, P["model.tgt_embed.0.lut.weight"]) * math.sqrt(d)
    return e + pe[pos0:pos0 + tok.size(1)]


def decode_tf(P, cfg, memory: Tensor, att_masks: Tensor, seq_in: Tensor, drop=None, rollouts: bool = False) -> Tensor:
    """Teacher-forced decoder (transformer.py:187-210); memory/att_masks already repeated per caption row.  Dropout sites as
    in `encode`, plus the PositionalEncoding's (transformer.py:398-401).  `rollouts`: the rows are SAMPLED captions whose
    log-probs the reference takes from its cached incremental passes (utils/training.py:224-237): those attend to every earlier
    position (transformer.py:265-269: mask = None once a cache exists), so the self-attention mask is the causal one alone —
    it differs from (seq != pad) & causal only behind a sampled token that carries the PAD id."""
    h = cfg.num_heads
    T = seq_in.size(1)
    x = _drop(drop, "emb", embed_tokens(P, cfg, seq_in))
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    keys = torch.ones_like(seq_in, dtype=torch.bool) if rollouts else (seq_in != cfg.pad_token_id)
    self_mask = (keys[:, None, :] & causal[None])[:, None]  # (R,1,T,T)
    src_mask = (att_masks != 0)[:, None, None, :]
    sa = getattr(cfg, "share_att_decoder", None)
    for l in range(cfg.num_layers):
        pre = f"model.decoder.layers.{l}."
        y = layer_norm(x, P[pre + "sublayer.0.norm.a_2"], P[pre + "sublayer.0.norm.b_2"])
        q, k, v, out = project_qkv(P, pre + "self_attn.", sa, y, y, h)
        x = x + _drop(drop, f"dec{l}.sub0", _linear(P, out, _merge(attention(q, k, v, self_mask, None, drop, f"dec{l}.self"))))
        y = layer_norm(x, P[pre + "sublayer.1.norm.a_2"], P[pre + "sublayer.1.norm.b_2"])
        q, k, v, out = project_qkv(P, pre + "src_attn.", sa, y, memory, h)
        x = x + _drop(drop, f"dec{l}.sub1", _linear(P, out, _merge(attention(q, k, v, src_mask, None, drop, f"dec{l}.cross"))))
        y = layer_norm(x, P[pre + "sublayer.2.norm.a_2"], P[pre + "sublayer.2.norm.b_2"])
        hid = _drop(drop, f"dec{l}.ffn", torch.relu(_linear(P, pre + "feed_forward.w_1", y)))
        x = x + _drop(drop, f"dec{l}.sub2", _linear(P, pre + "feed_forward.w_2", hid))
    return layer_norm(x, P["model.decoder.norm.a_2"], P["model.decoder.norm.b_2"])


def generator(P, x: Tensor) -> Tensor:
    """transformer.py:412-413."""
    return F.log_softmax(_linear(P, "model.generator.proj", x), dim=-1)


def forward_logp(P, cfg, att_feats, boxes, seqs, att_masks, drop=None, rollouts: bool = False) -> Tensor:
    """``_forward`` (relation_transformer.py:368-372): (R, T, V) log-probs, T = seqs.size(1)-1.  `drop`: see `_drop`; `rollouts`: see
    `decode_tf`."""
    mem = encode(P, cfg, att_feats, boxes, att_masks, drop)
    R, B = seqs.size(0), att_feats.size(0)
    if R != B:
        assert R % B == 0
        spi = R // B
        mem = mem.repeat_interleave(spi, 0)
        att_masks = att_masks.repeat_interleave(spi, 0)
    return generator(P, decode_tf(P, cfg, mem, att_masks, seqs[:, :-1], drop, rollouts))


def xe_loss(logp: Tensor, target: Tensor, mask: Tensor) -> Tensor:
    """LanguageModelCriterion, utils/losses.py:36-43; call with seqs[:,1:], masks[:,1:]."""
    target = target[:, : logp.size(1)]
    mask = mask[:, : logp.size(1)]
    out = -logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


def reward_loss(sample_logprobs: Tensor, seq: Tensor, reward: Tensor, pad: int = 0) -> Tensor:
    """RewardCriterion (utils/losses.py:15-29) with mask = seq != pad (utils/training.py:252-254)."""
    mask = (seq.reshape(-1, seq.size(-1)) != pad).float()
    out = -sample_logprobs.reshape(-1) * (mask * reward.reshape(-1, 1)).reshape(-1)
    return out.sum() / mask.sum()


# --------------------------------------------------------------------------------------------- incremental decode
class DecodeState:
    """KV caches for cached-attention decoding (transformer.py:240-273)."""

    def __init__(self, P, cfg, memory: Tensor, att_masks: Tensor):
        self.P, self.cfg = P, cfg
        self.memory, self.att_masks = memory, att_masks
        self.step = 0
        L = cfg.num_layers
        self.self_k = [None] * L
        self.self_v = [None] * L
        self.src_k = [None] * L
        self.src_v = [None] * L

    def repeat(self, n: int):
        """batch-repeat of every cache (transformer.py:240-252 + 489: repeat_interleave)."""
        r = lambda t: None if t is None else t.repeat_interleave(n, 0)
        self.memory, self.att_masks = r(self.memory), r(self.att_masks)
        for lst in (self.self_k, self.self_v, self.src_k, self.src_v):
            for i in range(len(lst)):
                lst[i] = r(lst[i])

    def reorder(self, idx: Tensor):
        for lst in (self.self_k, self.self_v, self.src_k, self.src_v):
            for i in range(len(lst)):
                if lst[i] is not None:
                    lst[i] = lst[i][idx]


def decode_step(st: DecodeState, it: Tensor, drop=None) -> Tensor:
    """One ``get_logprobs_state`` call (relation_transformer.py:374-387): tokens (rows,) -> logp (rows,V).
    `drop` (see `_drop`; train-mode sampling, utils/training.py:224-237): called as drop(site, x) with the sites of `decode_tf`
    and the tensors of THIS position ((rows, 1, d), probabilities (rows, h, 1, keys so far)); it reads the position from
    `st.step - 1`."""
    P, cfg = st.P, st.cfg
    h = cfg.num_heads
    x = embed_tokens(P, cfg, it[:, None], pos0=st.step)
    st.step += 1
    x = _drop(drop, "emb", x)
    src_mask = (st.att_masks != 0)[:, None, None, :]
    sa = getattr(cfg, "share_att_decoder", None)
    for l in range(cfg.num_layers):
        pre = f"model.decoder.layers.{l}."
        y = layer_norm(x, P[pre + "sublayer.0.norm.a_2"], P[pre + "sublayer.0.norm.b_2"])
        q, k, v, out = project_qkv(P, pre + "self_attn.", sa, y, y, h)
        if st.self_k[l] is not None:
            k = torch.cat((st.self_k[l], k), 2)
            v = torch.cat((st.self_v[l], v), 2)
        st.self_k[l], st.self_v[l] = k, v
        x = x + _drop(drop, f"dec{l}.sub0", _linear(P, out, _merge(attention(q, k, v, None, None, drop, f"dec{l}.self"))))
        y = layer_norm(x, P[pre + "sublayer.1.norm.a_2"], P[pre + "sublayer.1.norm.b_2"])
        q, k, v, out = project_qkv(P, pre + "src_attn.", sa, y, st.memory, h, need_kv=st.src_k[l] is None)
        if st.src_k[l] is None:
            st.src_k[l], st.src_v[l] = k, v
        x = x + _drop(drop, f"dec{l}.sub1", _linear(P, out, _merge(attention(q, st.src_k[l], st.src_v[l], src_mask, None, drop, f"dec{l}.cross"))))
        y = layer_norm(x, P[pre + "sublayer.2.norm.a_2"], P[pre + "sublayer.2.norm.b_2"])
        hid = _drop(drop, f"dec{l}.ffn", torch.relu(_linear(P, pre + "feed_forward.w_1", y)))
        x = x + _drop(drop, f"dec{l}.sub2", _linear(P, pre + "feed_forward.w_2", hid))
    x = layer_norm(x, P["model.decoder.norm.a_2"], P["model.decoder.norm.b_2"])
    return generator(P, x[:, -1])


def gumbel_from_hash(seed: int, t: int, rows: int, vocab: int, row_offset: int = 0) -> Tensor:
    """Counter-based uniforms -> Gumbel noise; the SAME integer hash runs inside the HIP sampler
    (csrc/ortk_common.h: ortk_hash_u32), so multinomial decoding is reproducible token-for-token.  `row_offset`: the draws are
    keyed by the GLOBAL row (ortk_decode_opts.sample_row_offset: a data-parallel rank samples what one process samples on the
    whole batch)."""
    r = torch.arange(row_offset, row_offset + rows, dtype=torch.int64)[:, None]
    v = torch.arange(vocab, dtype=torch.int64)[None, :]
    M = 0xFFFFFFFF
    x = (r * 0x9E3779B1 + v * 0x85EBCA77 + (t + 1) * 0xC2B2AE3D + seed * 0x27D4EB2F) & M
    x ^= x >> 16; x = (x * 0x7FEB352D) & M
    x ^= x >> 15; x = (x * 0x846CA68B) & M
    x ^= x >> 16
    u = ((x >> 8).to(torch.float32) + 0.5) * (1.0 / 16777216.0)
    return -torch.log(-torch.log(u))


def sample_greedy_or_multinomial(P, cfg, att_feats, boxes, att_masks, num_random_sample: int = 0,
                                 temperature: float = 1.0, decoding_constraint: int = 0, seed: int = 0, drop=None, drop_step=None,
                                 scores_out=None, sample_row_offset: int = 0):
    """``_generate_captions`` greedy / multinomial branches (transformer.py:507-561).

    Multinomial draws use Gumbel-max over ``gumbel_from_hash`` (an exact sampler of
    multinomial(exp(logp/temperature))); the reference's torch RNG stream cannot be reproduced.
    """
    L = cfg.max_seq_length
    mem = encode(P, cfg, att_feats, boxes, att_masks, drop)          # (drop: encoder sites; drop_step(t): the decoder sites of position t)
    n = att_feats.size(0)
    if num_random_sample > 0:
        mem = mem.repeat_interleave(num_random_sample, 0)
        att_masks = att_masks.repeat_interleave(num_random_sample, 0)
        n *= num_random_sample
    st = DecodeState(P, cfg, mem, att_masks)
    it = torch.full((n,), cfg.bos_token_id, dtype=torch.long)
    seq = torch.zeros(n, L, dtype=torch.long)
    seq_lp = torch.zeros(n, L)
    unfinished = it != cfg.eos_token_id
    for t in range(L):
        logp = decode_step(st, it, None if drop_step is None else drop_step(t))
        if decoding_constraint and t > 0:
            logp = logp.scatter(1, seq[:, t - 1:t], float("-inf"))
        if num_random_sample > 0:
            z = logp / temperature + gumbel_from_hash(seed, t, n, logp.size(1), sample_row_offset)
            if scores_out is not None:          # (tests: the perturbed scores of every step, to show that a flipped token is a near-tie)
                scores_out.append(z.clone())
            it = z.argmax(-1)
            lp = logp.gather(1, it[:, None]).squeeze(1)
        else:
            lp, it = logp.max(1)
        seq[:, t] = it * unfinished.long()
        unfinished = unfinished & (it != cfg.eos_token_id)
        seq_lp[:, t] = lp
        if unfinished.sum() == 0:
            break
    k = max(1, num_random_sample)
    return seq.view(-1, k, L), seq_lp.view(-1, k, L)


def _length_penalty(cfgstr: str):
    """utils/model_utils.py:121-146."""
    if cfgstr == "":
        return lambda length, lp: lp
    kind, alpha = cfgstr.split("_")
    alpha = float(alpha)
    if kind == "wu":
        return lambda length, lp: lp / (((5 + length) ** alpha) / ((5 + 1) ** alpha))
    if kind == "avg":
        return lambda length, lp: lp / length
    raise ValueError(cfgstr)


def beam_search(P, cfg, att_feats, boxes, att_masks, beam_size: int, temperature: float = 1.0,
                decoding_constraint: int = 0, length_penalty: str = ""):
    """Beam branch of ``_generate_captions`` (transformer.py:481-505) + ``batch_beam_search`` with
    group_size 1 (caption_model.py:56-111,151-226).  Returns seq (N,b,L), seq_logprobs (N,b,L), p (N,b)."""
    L, V, b = cfg.max_seq_length, cfg.vocab_size, beam_size
    pen = _length_penalty(length_penalty)
    mem = encode(P, cfg, att_feats, boxes, att_masks)
    N = att_feats.size(0)
    st = DecodeState(P, cfg, mem, att_masks)
    logp = decode_step(st, torch.full((N,), cfg.bos_token_id, dtype=torch.long))  # (N,V)
    st.repeat(b)
    beam_seq = torch.zeros(N, b, 0, dtype=torch.long)
    beam_tok_lp = torch.zeros(N, b, 0)  # log-prob of each chosen token (gathered from the un-augmented logp)
    cum = torch.zeros(N, b)
    done = [[] for _ in range(N)]
    for t in range(L):
        if decoding_constraint and t > 0:
            logp = logp.scatter(1, beam_seq[:, :, t - 1].reshape(-1, 1), float("-inf"))
        lp3 = logp.reshape(N, -1, V)
        cand = (cum[:, :1] if t == 0 else cum).unsqueeze(-1) + lp3  # (N,q,V)
        ys, ix = torch.sort(cand.reshape(N, -1), -1, True)
        ys, ix = ys[:, :b], ix[:, :b]
        parent = ix // V
        tok = ix % V
        if t > 0:
            beam_seq = beam_seq.gather(1, parent[:, :, None].expand_as(beam_seq))
            beam_tok_lp = beam_tok_lp.gather(1, parent[:, :, None].expand_as(beam_tok_lp))
        beam_seq = torch.cat([beam_seq, tok[:, :, None]], -1)
        beam_tok_lp = torch.cat([beam_tok_lp, lp3.reshape(N, -1).gather(1, ix)[:, :, None]], -1)
        cum = ys.clone()
        state_ix = (parent + torch.arange(N)[:, None] * lp3.size(1)).reshape(-1)
        if t > 0:
            st.reorder(state_ix)
        is_end = tok == cfg.eos_token_id
        if t == L - 1:
            is_end = torch.ones_like(is_end)
        for n in range(N):
            for q in range(b):
                if is_end[n, q]:
                    done[n].append(dict(seq=beam_seq[n, q].clone(), lps=beam_tok_lp[n, q].clone(),
                                        p=pen(t + 1, cum[n, q].item())))
        cum = cum - 1000.0 * is_end.float()
        if t < L - 1:  # the reference runs one more (unused) decoder pass after the last step
            logp = decode_step(st, tok.reshape(-1))
            logp = F.log_softmax(logp / temperature, dim=-1)
    seq = torch.zeros(N, b, L, dtype=torch.long)
    seq_lp = torch.zeros(N, b, L)
    ps = torch.zeros(N, b)
    for n in range(N):
        best = sorted(done[n], key=lambda d: -d["p"])[:b]  # python sort is stable, as in the reference
        for q, d in enumerate(best):
            ln = d["seq"].numel()
            seq[n, q, :ln] = d["seq"]
            seq_lp[n, q, :ln] = d["lps"]
            ps[n, q] = d["p"]
    return seq, seq_lp, ps


# --------------------------------------------------------------------------------------------- pruning
def is_mask(name: str) -> bool:
    return name.endswith("_pruning_mask")


class _STE(torch.autograd.Function):
    """Straight-through sample (pruning/sampler.py:10-34): forward = given sample, backward = identity."""

    @staticmethod
    def forward(ctx, probs, sample):
        return sample

    @staticmethod
    def backward(ctx, g):
        return g, None


def round_sigmoid(m: Tensor) -> Tensor:
    """pruning/sampler.py:27-34,57-66: torch.round is round-half-to-even (logit 0 -> pruned); the backward is
    straight-through over the real sigmoid derivative, so the sparsity loss DOES reach the mask logits."""
    probs = torch.sigmoid(m)
    return _STE.apply(probs, torch.round(probs.detach()))


def effective_params(P: Dict[str, Tensor], mask_type: str, training: bool = False,
                     samples: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """``MaskMixin.get_masked_weight`` for every masked tensor (pruning/masked_layer.py:84-110).

    supermask: eval -> round(sigmoid(m)); train -> the provided Bernoulli ``samples[name]`` passed
    straight-through over sigmoid(m).  Other mask types: the stored binary mask itself.
    Returns a dict with the dense-model key names holding ``s * W``.
    """
    out = {}
    for name, w in P.items():
        if is_mask(name):
            continue
        mname = name + "_pruning_mask"
        if mname not in P:
            out[name] = w
            continue
        m = P[mname]
        if mask_type == "supermask":
            probs = torch.sigmoid(m)
            s = _STE.apply(probs, samples[mname] if training else torch.round(probs.detach()))
        else:
            s = m
        out[name] = s * w
    return out


def mask_sparsities(P, mask_type: str, names: Optional[Sequence[str]] = None):
    """``all_mask_sparsities`` (pruning/prune.py:124-144): total sparsity, nnz, per-tensor sparsity."""
    names = [n for n in P if is_mask(n)] if names is None else list(names)
    ms = [round_sigmoid(P[n]) if mask_type == "supermask" else P[n] for n in names]
    nel = [m.nelement() for m in ms]
    nnz = [m.sum() for m in ms]
    per = [1.0 - (z / n) for z, n in zip(nnz, nel)]
    tot_nnz = sum(nnz)
    return 1.0 - (tot_nnz / sum(nel)), tot_nnz, per, names


def sparsity_loss(P, target: float, weight: float, step: int, max_step: int) -> Tensor:
    """``compute_sparsity_loss`` (pruning/prune.py:228-269)."""
    total, _, _, _ = mask_sparsities(P, "supermask")
    loss = torch.abs(target - total)
    s = 1.0 + torch.cos(torch.tensor(min(1.0, step / max_step) * math.pi))
    return loss * weight * (1.0 - s / 2)


def compute_mask(criterion: Tensor, sparsity_target: float) -> Tensor:
    """pruning/prune.py:271-283: zero the ``int(target*numel)`` smallest entries."""
    mask = torch.ones_like(criterion)
    k = int(sparsity_target * criterion.nelement())
    if k > 0:
        idx = torch.topk(criterion.view(-1), k=k, largest=False).indices
        mask.view(-1)[idx] = 0
    return mask


def update_masks_once(P, mask_type: str, sparsity_target: float, mask_grads: Optional[Dict[str, Tensor]] = None):
    """pruning/prune.py:296-373.  Returns {mask_name: new binary mask}."""
    mnames = [n for n in P if is_mask(n)]
    weights = [P[n[: -len("_pruning_mask")]] for n in mnames]
    if mask_type == "snip":
        vec = torch.cat([mask_grads[n].reshape(-1) for n in mnames])
        crit = [vec / vec.sum()]
    elif mask_type in ("mag_dist", "mag_grad_dist", "lottery_mag_dist"):
        cs = []
        for w in weights:
            sd = torch.std(w.reshape(-1), dim=0, unbiased=False)
            cs.append(torch.abs((w - w.mean()) / sd).reshape(-1))
        crit = [torch.cat(cs)]
    elif mask_type in ("mag_uniform", "mag_grad_uniform", "lottery_mag_uniform"):
        crit = [torch.abs(w) for w in weights]
    elif mask_type in ("mag_blind", "mag_grad_blind", "lottery_mag_blind"):
        crit = [torch.cat([torch.abs(w).reshape(-1) for w in weights])]
    else:
        raise ValueError(mask_type)
    new = [compute_mask(c, sparsity_target) for c in crit]
    if len(new) == 1:
        new = torch.split(new[0], [w.nelement() for w in weights])
    return {n: m.reshape(P[n].shape) for n, m in zip(mnames, new)}


def gradual_sparsity(target: float, step: int, start: int, prune_steps: int, initial: float = 0.0, freq: int = 1000):
    """Cubic schedule of ``update_masks_gradual`` (pruning/prune.py:375-433); None when not a pruning step."""
    t0, tn = start, start + freq * prune_steps
    if not (step >= t0 and (step <= tn or tn < 0) and (step - t0) % freq == 0):
        return None
    p = min(1.0, max(0.0, (step - t0) / (tn - t0)))
    return target + (initial - target) * ((1.0 - p) ** 3)


# --------------------------------------------------------------------------------------------- optimiser
def noam_rate(step: int, d_model: int, factor: float, warmup: int) -> float:
    """utils/optim.py:46-49."""
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


def adam_clip_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: dict, lr: float,
                   clip: float = 0.1, betas=(0.9, 0.98), eps: float = 1e-9):
    """clip_grad_value_ (utils/optim.py:187-191) then torch.optim.Adam's update rule (utils/optim.py:116-126)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for n, p in params.items():
        g = grads[n].clamp(-clip, clip)
        m = state.setdefault("m/" + n, torch.zeros_like(p))
        v = state.setdefault("v/" + n, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
    return params
