"""Batch assembly for the path: the reference's ``UpDownCollate`` / ``ObjectRelationCollate``
(``sparse_caption/data/collate.py:77-227``) with the same constructor, ``__call__(batch)`` contract, config attributes,
argparse options and output dict — ``att_feats (B, Smax, F)``, ``att_masks (B, Smax)``, ``boxes (B, Smax, 4)``, ``seqs (R, T)``
int64, ``masks (R, T)``, ``gts``, ``image_paths``, ``image_ids`` — so that a ``DataLoader(collate_fn=...)`` of the reference
feeds this package's models unchanged.

What is different: the zero-padding of the ragged per-image arrays (10-100 detected regions) is one multi-threaded native
call (``ortk_pad_rows`` in ``libortk.so``, ``include/ortk_data.h``) into a PINNED buffer when a GPU is present, so that
``batch[k].cuda(non_blocking=True)`` overlaps with compute; the reference pads with ``torch.nn.utils.rnn.pad_sequence`` on
pageable memory.  Reading the ``.npy`` files, the optional multiprocessing cache, the caption sampling (``random.sample``)
and the tokenizer call are the reference's steps in the reference's order (so a seeded ``random`` gives the same batch).
The tokenizer is whatever object the caller passes (``encode(text, add_bos_eos=True, max_seq_length=...)``): tokenisation
is outside this package's scope.
"""
import ctypes as C
import logging
import os
import random

import numpy as np
import torch

from .. import _lib as L

logger = logging.getLogger(__name__)

_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
_SIG = {
    "ortk_pad_rows": (_I32, [_P, _P, _I64, _I64, _I64, _P, _P, _I32]),
    "ortk_pad_seqs": (_I32, [_P, _P, _I64, _I64, _I64, _P, _P]),
}


def _lib():
    lib = L.lib()
    if not getattr(lib, "_data_bound", False):
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib._data_bound = True
    return lib


def _empty(shape, dtype):
    """Pinned when a GPU is there to copy to (the batch's only consumer), pageable otherwise."""
    pin = torch.cuda.device_count() > 0
    try:
        return torch.empty(shape, dtype=dtype, pin_memory=pin)
    except RuntimeError:
        return torch.empty(shape, dtype=dtype)


def pad_rows(arrays, want_mask=False, nthreads=0):
    """list of (n_i, F) float32 arrays -> (B, max n_i, F) float32 (zeros behind each image's rows) [, (B, max n_i) mask]."""
    arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    B = len(arrays)
    F = arrays[0].shape[1] if B else 1
    assert all(a.ndim == 2 and a.shape[1] == F for a in arrays), "every array must be (n_i, F) with one F"
    smax = max((a.shape[0] for a in arrays), default=0)
    out = _empty((B, smax, F), torch.float32)
    mask = _empty((B, smax), torch.float32) if want_mask else None
    ptrs = (C.c_void_p * max(B, 1))(*[a.ctypes.data for a in arrays])
    n = np.array([a.shape[0] for a in arrays], dtype=np.int64)
    rc = _lib().ortk_pad_rows(C.cast(ptrs, _P), n.ctypes.data_as(_P), B, F, smax, out.data_ptr(),
                              mask.data_ptr() if want_mask else None, nthreads)
    if rc != 0:
        raise ValueError("ortk_pad_rows: bad arguments")
    return (out, mask) if want_mask else out


def pad_seqs(seqs, pad=0):
    """list of 1-D int64 token-id sequences -> (R, max len) int64 padded with `pad`, and the (R, max len) float mask."""
    seqs = [np.ascontiguousarray(np.asarray(s), dtype=np.int64).reshape(-1) for s in seqs]
    R = len(seqs)
    smax = max((s.shape[0] for s in seqs), default=0)
    out = _empty((R, smax), torch.int64)
    mask = _empty((R, smax), torch.float32)
    ptrs = (C.c_void_p * max(R, 1))(*[s.ctypes.data for s in seqs])
    n = np.array([s.shape[0] for s in seqs], dtype=np.int64)
    rc = _lib().ortk_pad_seqs(C.cast(ptrs, _P), n.ctypes.data_as(_P), R, smax, pad, out.data_ptr(), mask.data_ptr())
    if rc != 0:
        raise ValueError("ortk_pad_seqs: bad arguments")
    return out, mask


class ListDataset(torch.utils.data.Dataset):
    """Basically a `list` (collate.py:31-41)."""

    def __init__(self, data):
        self.data = data

    def __getitem__(self, index):
        return self.data[index]

    def __len__(self):
        return len(self.data)


class UpDownCollate:
    """collate.py:77-188."""

    def __init__(self, config, tokenizer, cache_dict=None):
        self.config = config
        self.tokenizer = tokenizer
        import multiprocessing.managers as mp
        self.cache_dict = cache_dict if isinstance(cache_dict, mp.DictProxy) else None
        if self.cache_dict is not None:
            logger.info(f"{self.__class__.__name__}: Using multiprocessing cache dict.")
        if self.config.input_att_dir is None:
            self.config.input_att_dir = self.join_default_bu_dir("cocobu_att")
        assert self.config.seq_per_img > 0, "`self.config.seq_per_img` should be greater than 0"

    def join_default_bu_dir(self, dirname):
        return os.path.join(self.config.dataset_dir, "bu", dirname)

    def _cache_data(self, key, key_value_fn):
        if self.cache_dict is None:
            return key_value_fn(key)
        try:
            data = self.cache_dict[key]
        except KeyError:
            data = key_value_fn(key)
            try:
                import psutil
                vm = psutil.virtual_memory()
                free = vm.available / vm.total
            except Exception:   # pragma: no cover
                free = 1.0
            if free > max(0.2, getattr(self.config, "cache_min_free_ram", 0.2)):
                self.cache_dict[key] = data
        return data

    @staticmethod
    def _get_att_feats(path):
        data = np.load(path)
        return data.reshape(-1, data.shape[-1]).astype("float32")

    def __call__(self, batch):
        config = self.config
        image_paths, image_ids, captions, all_captions, all_gts = zip(*batch)
        att_feats = [self._cache_data(os.path.join(config.input_att_dir, f"{imgid}.npy"), self._get_att_feats) for imgid in image_ids]
        labels = [
            np.asarray(self.tokenizer.encode(_, add_bos_eos=True, max_seq_length=config.max_seq_length), dtype=np.int64)
            for gt in all_captions
            for _ in random.sample(gt, min(config.seq_per_img, len(gt)))
        ]
        feats, att_masks = pad_rows(att_feats, want_mask=True)
        seqs, masks = pad_seqs(labels, 0)
        return {"att_feats": feats, "att_masks": att_masks, "seqs": seqs, "masks": masks, "gts": all_gts,
                "image_paths": image_paths, "image_ids": image_ids}

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--max_seq_length", type=int, default=18, help="int: Maximum sequence length including <BOS> and <EOS>.")
        parser.add_argument("--seq_per_img", type=int, default=5, help="Number of captions to sample for each image during training.")
        parser.add_argument("--input_att_dir", type=str, default=None,
                            help="str: path to the directory containing the preprocessed att feats")


class ObjectRelationCollate(UpDownCollate):
    """collate.py:191-227."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.config.input_rel_box_dir is None:
            self.config.input_rel_box_dir = self.join_default_bu_dir("cocobu_box_relative")

    @staticmethod
    def _get_boxes(path):
        return np.load(path).astype("float32")

    def __call__(self, batch):
        config = self.config
        image_ids = list(zip(*batch))[1]
        data = super().__call__(batch)
        boxes = [self._cache_data(os.path.join(config.input_rel_box_dir, f"{imgid}.npy"), self._get_boxes) for imgid in image_ids]
        data["boxes"] = pad_rows(boxes)
        return data

    @staticmethod
    def add_argparse_args(parser):
        UpDownCollate.add_argparse_args(parser)
        parser.add_argument("--input_rel_box_dir", type=str, default=None,
                            help="str: this directory contains the bounding boxes in relative coordinates "
                                 "for the corresponding image features in --input_att_dir")
