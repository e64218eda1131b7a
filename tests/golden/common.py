"""Shared, reference-free helpers for the golden fixtures.

Everything here is plain numpy so that the SAME deterministic weights / inputs can be rebuilt
(a) inside ``make_golden.py`` (which imports the reference in the build container) and
(b) inside the tests on the GPU box (where ``/root/reference`` does not exist).

Weights come from a frozen ``numpy.random.RandomState`` stream, keyed by the parameter NAME, so the
values do not depend on module registration order (the state_dict key names are the drop-in
contract).
"""
import zlib
import numpy as np

PAD, UNK, BOS, EOS = 0, 1, 2, 3

TINY_CFG = dict(
    d_model=64, dim_feedforward=128, num_layers=2, num_heads=8, drop_prob_src=0.5, max_seq_length=18,
    att_feat_size=96, vocab_size=101, bos_token_id=BOS, eos_token_id=EOS, unk_token_id=UNK, pad_token_id=PAD,
    share_att_encoder=None, share_att_decoder=None, share_layer_encoder=None, share_layer_decoder=None,
    no_box_trigonometric_embedding=False,
    prune_type="supermask", prune_mask_freeze_scope="", prune_supermask_init=5.0,
)

FULL_CFG = dict(
    d_model=512, dim_feedforward=2048, num_layers=6, num_heads=8, drop_prob_src=0.5, max_seq_length=18,
    att_feat_size=2048, vocab_size=10001, bos_token_id=BOS, eos_token_id=EOS, unk_token_id=UNK, pad_token_id=PAD,
    share_att_encoder=None, share_att_decoder=None, share_layer_encoder=None, share_layer_decoder=None,
    no_box_trigonometric_embedding=False,
    prune_type="supermask", prune_mask_freeze_scope="", prune_supermask_init=5.0,
)


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def make_param(name: str, shape, seed: int, gen_scale: float = 1.0, eos_bias: float = 0.0) -> np.ndarray:
    """Deterministic value for one state_dict entry (fp32)."""
    rs = _rs(name, seed)
    shape = tuple(int(s) for s in shape)
    if name.endswith("_pruning_mask"):
        raise ValueError("mask logits are produced by make_mask_logits")
    if len(shape) >= 2:
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        a = np.sqrt(6.0 / (fan_in + fan_out))
        w = rs.uniform(-a, a, size=shape)
        if name.endswith("generator.proj.weight"):
            w = w * gen_scale
        if ".WGs." in name:
            # keep a healthy share of positive pre-activations so relu(WG e + b) is not all-zero
            w = w * 2.0
        return w.astype(np.float32)
    if name.endswith(".a_2"):
        return (1.0 + rs.uniform(-0.2, 0.2, size=shape)).astype(np.float32)
    v = rs.uniform(-0.1, 0.1, size=shape)
    if ".WGs." in name:  # bias of the geometry linears: push positive so the bias matters
        v = v + 0.3
    if name.endswith("generator.proj.bias") and eos_bias != 0.0:
        v[EOS] += eos_bias
    return v.astype(np.float32)


def make_mask_logits(name: str, shape, seed: int, keep_prob: float) -> np.ndarray:
    """Supermask logits with round(sigmoid(m)) ~ Bernoulli(keep_prob); values kept away from 0."""
    rs = _rs(name + "#mask", seed)
    keep = rs.uniform(size=shape) < keep_prob
    mag = rs.uniform(0.5, 4.0, size=shape)
    return np.where(keep, mag, -mag).astype(np.float32)


def make_inputs(seed: int, n_img: int, n_reg: int, feat: int, vocab: int, spi: int, seq_len: int = 18,
                ragged: bool = True):
    """Synthetic batch in ObjectRelationCollate layout (reference data/collate.py:119-169,202-216)."""
    rs = np.random.RandomState(seed)
    att = rs.gamma(0.5, 2.3, size=(n_img, n_reg, feat)) * (rs.uniform(size=(n_img, n_reg, feat)) < 0.7)
    x0 = rs.uniform(0, 0.7, size=(n_img, n_reg)); y0 = rs.uniform(0, 0.7, size=(n_img, n_reg))
    w = rs.uniform(0.03, 0.6, size=(n_img, n_reg)); h = rs.uniform(0.03, 0.6, size=(n_img, n_reg))
    boxes = np.stack([x0, y0, np.minimum(x0 + w, 1.0), np.minimum(y0 + h, 1.0)], -1)
    masks = np.ones((n_img, n_reg), np.float32)
    if ragged and n_img > 1:
        for i in range(1, n_img):
            n_valid = int(rs.randint(max(1, n_reg // 3), n_reg + 1))
            masks[i, n_valid:] = 0
        masks[0, :] = 1  # batch max == padded length (collate pads to the batch max)
    att = att * masks[..., None]
    boxes = boxes * masks[..., None]
    R = n_img * spi
    seqs = np.zeros((R, seq_len), np.int64)
    smask = np.zeros((R, seq_len), np.float32)
    for r in range(R):
        L = int(rs.randint(3, seq_len - 1))  # words, so BOS + L + EOS <= seq_len
        seqs[r, 0] = BOS
        seqs[r, 1:1 + L] = rs.randint(4, vocab, size=L)
        seqs[r, 1 + L] = EOS
        smask[r, :L + 2] = 1
    return dict(att_feats=att.astype(np.float32), boxes=boxes.astype(np.float32), att_masks=masks,
                seqs=seqs, masks=smask)


def state_dict_from_shapes(shapes: dict, seed: int, gen_scale=1.0, eos_bias=0.0, keep_prob=None) -> dict:
    """shapes: name -> shape for every parameter (no buffers). Returns name -> np.float32 array."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith("_pruning_mask"):
            out[name] = make_mask_logits(name, shape, seed, keep_prob if keep_prob is not None else 0.5)
        else:
            out[name] = make_param(name, shape, seed, gen_scale, eos_bias)
    return out


# ---- fixture recipes (shared by make_golden.py and the tests) ----
G1_SEED, G1_GEN_SCALE, G1_EOS_BIAS = 1234, 3.0, 3.2
G1_INPUTS = dict(seed=77, n_img=3, n_reg=12, feat=TINY_CFG["att_feat_size"], vocab=TINY_CFG["vocab_size"], spi=2)
G2_SEED = 8888
G2_INPUTS = dict(seed=99, n_img=4, n_reg=36, feat=2048, vocab=10001, spi=5, ragged=False)
G3_KEEP = 0.3


# ---- collate fixture (G12): synthetic feature / box files + a stand-in tokenizer, shared by make_golden_collate.py and the tests ----
class StubTokenizer:
    """Stands in for the reference's tokenizers (out of scope): words -> ids by a fixed arithmetic rule."""

    def __init__(self, vocab=97, bos=2, eos=3):
        self.vocab, self.bos, self.eos = vocab, bos, eos

    def encode(self, text, add_bos_eos=True, max_seq_length=18):
        ids = [4 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.vocab - 4)) for w in text.split()]
        if add_bos_eos:
            ids = [self.bos] + ids[:max_seq_length - 2] + [self.eos]
        return ids[:max_seq_length]


COLLATE_WORDS = "a an the man woman dog cat sits stands on in near table street grass with red blue small large ball hat".split()


def make_collate_fixture(root, seed=4242, n_img=5, feat=16):
    """Writes <root>/att/<id>.npy ((n, feat) or (1, n, feat)) and <root>/box/<id>.npy ((n, 4)) with n in 3..12 and returns the
    list of dataset items (image_path, image_id, caption, all_captions, all_gts) a reference DataLoader would hand to the collate."""
    import os
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "att"), exist_ok=True)
    os.makedirs(os.path.join(root, "box"), exist_ok=True)
    items = []
    for i in range(n_img):
        n = int(rs.randint(3, 13))
        att = rs.standard_normal((n, feat)).astype(np.float32)
        if i % 2:
            att = att[None]                                     # some files carry a leading axis (collate.py:109-110 reshapes)
        np.save(os.path.join(root, "att", f"{100 + i}.npy"), att)
        np.save(os.path.join(root, "box", f"{100 + i}.npy"), rs.uniform(0, 1, (n, 4)).astype(np.float64 if i == 2 else np.float32))
        ncap = int(rs.randint(1, 5))
        caps = [" ".join(rs.choice(COLLATE_WORDS, size=int(rs.randint(3, 25)))) for _ in range(ncap)]
        items.append((f"/images/{100 + i}.jpg", 100 + i, caps[0], caps, [c.split() for c in caps]))
    return items
