"""The reference's plain ``transformer`` model (``sparse_caption/models/transformer.py:617-719``) on the same HIP path: the
relation transformer without the geometry bias.  Differences the executor honours (``ortk_config.no_box``):

* encoder self-attention is plain scaled-dot-product attention (no ``WGs`` parameters, no boxes input);
* ``core.src_embed`` = Linear + ReLU + Dropout on EVERY region row (the relation transformer zeroes padded regions,
  model_utils.py:149-168; here they are only masked as attention keys, transformer.py:78-81);
* state_dict names of that class: ``core.src_embed.0.*``, ``core.encoder.*``, ``core.decoder.*``, ``core.tgt_embed.0.lut.weight``,
  ``core.tgt_embed.1.pe``, ``core.generator.proj.*``; xavier-uniform on every >= 2-D parameter (transformer.py:660-664).

Call contract (transformer.py:666-676,705-710): ``model(att_feats=, att_masks=, seqs=)`` -> log-probs, ``mode="sample"`` with
``opt``; extra keys of the batch dict (``boxes`` from an object-relation collate) are ignored.
"""
import torch

from . import register_model
from ..data.collate import UpDownCollate
from .relation_transformer import RelationTransformerModel


@register_model("transformer")
class TransformerModel(RelationTransformerModel):
    COLLATE_FN = UpDownCollate     # no boxes in the batch dict (data/collate.py:77-188)
    NO_BOX = True

    def _prepare(self, att_feats, boxes, att_masks, att_max_len=None):
        # the executor never reads the boxes of this model; a (B, S, 4) placeholder keeps the shared batch plumbing
        boxes = att_feats.new_zeros(att_feats.shape[0], att_feats.shape[1], 4)
        return super()._prepare(att_feats, boxes, att_masks, att_max_len)

    def _forward(self, att_feats, att_masks=None, seqs=None, boxes=None, **kwargs):
        return super()._forward(att_feats, None, seqs, att_masks, **kwargs)

    def _sample(self, att_feats, att_masks=None, opt=None, boxes=None, **kwargs):
        return super()._sample(att_feats, None, att_masks, opt)

    def encode(self, att_feats, att_masks=None, boxes=None):
        return super().encode(att_feats, None, att_masks)
