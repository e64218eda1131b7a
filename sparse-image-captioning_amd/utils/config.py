"""Attribute bag with ``.get`` / ``.update`` — the part of the reference's ``utils/config.py:20-182`` that the
models read (``config.<attr>``, ``opt.get(key, default)``)."""
import json
from copy import deepcopy


class Config:
    def __init__(self, x: str = None, **kwargs):
        if x is not None:
            if not isinstance(x, str):
                raise TypeError(f"Positional argument must be a string, saw {type(x)}")
            kwargs.update(json.loads(x))
        for key, value in kwargs.items():
            setattr(self, key, value)

    def get(self, key, default_value=None):
        return vars(self).get(key, default_value)

    def update(self, kv_mapping):
        return vars(self).update(kv_mapping)

    def dict(self):
        return dict(vars(self))

    def json(self, **kwargs):
        kwargs.setdefault("indent", 2)
        kwargs.setdefault("sort_keys", True)
        return json.dumps(self.dict(), **kwargs)

    def deepcopy(self):
        return deepcopy(self)

    def __repr__(self):
        return self.json()


# defaults of the model flags (models/transformer.py:563-614, relation_transformer.py:414-426, pruning/prune.py:435-476)
ORT_DEFAULTS = dict(
    d_model=512, dim_feedforward=2048, num_layers=6, num_heads=8, drop_prob_src=0.5, max_seq_length=18,
    att_feat_size=2048, vocab_size=10001, bos_token_id=2, eos_token_id=3, unk_token_id=1, pad_token_id=0,
    share_att_encoder=None, share_att_decoder=None, share_layer_encoder=None, share_layer_decoder=None,
    no_box_trigonometric_embedding=False, prune_type="supermask", prune_mask_freeze_scope="",
    prune_supermask_init=5.0, seq_per_img=5,
)


def ort_config(**overrides):
    d = dict(ORT_DEFAULTS)
    d.update(overrides)
    return Config(**d)
