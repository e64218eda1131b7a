"""Checkpoint helpers mirrored from ``sparse_caption/utils/model_utils.py:105-118``."""
import torch


def count_nonzero(tensor):
    """model_utils.py:105-106."""
    return tensor.ne(0).float().sum()


def densify_state_dict(state_dict):
    """COO-sparse (``state_dict_sparse``, optionally fp16: scripts/train_n_prune_transformer.py:251-291) -> dense fp32-or-
    as-stored tensors (model_utils.py:109-118).  The HIP models accept sparse / half entries directly in
    ``load_state_dict`` (they pass through this function and an fp32 cast)."""
    return {k: (v.to_dense() if isinstance(v, torch.Tensor) and v.is_sparse else v) for k, v in state_dict.items()}
