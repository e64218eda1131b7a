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
    """Pinned when a GPU is there to copy to AND this is the main process; pageable inside a DataLoader worker (the batch is
    copied into shared memory on its way back, so pinning there is lost work — use DataLoader(pin_memory=True) instead)."""
    pin = torch.cuda.device_count() > 0 and torch.utils.data.get_worker_info() is None
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
    """A list behind the Dataset protocol (what the reference's loaders wrap, collate.py:31-41)."""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return self.data[index]


def _load_features(path):
    """(..., F) array on disk -> (regions, F) float32."""
    arr = np.load(path)
    return np.asarray(arr, dtype=np.float32).reshape(-1, arr.shape[-1])


def _load_boxes(path):
    return np.asarray(np.load(path), dtype=np.float32)


class _RegionBatcher:
    """Shared machinery of the two collate classes below.

    A subclass lists its per-image region arrays in ``SOURCES``: ``(batch key, config attribute holding the directory,
    default sub-directory of <dataset_dir>/bu, loader, emit a validity mask?)``.  One call turns a list of dataset items
    ``(image_path, image_id, caption, all_captions, gts)`` into the batch dict of the reference's collate functions
    (data/collate.py:119-169, 202-216): for every source the arrays ``<dir>/<image_id>.npy`` zero-padded to the longest image
    of the batch, plus ``seq_per_img`` captions per image drawn with ``random.sample`` in item order (same draws as the
    reference for a seeded ``random``), tokenised and padded with 0."""

    SOURCES = ()

    def __init__(self, config, tokenizer, cache_dict=None):
        self.config, self.tokenizer = config, tokenizer
        from multiprocessing.managers import DictProxy
        # only a Manager().dict() is shared between DataLoader workers; anything else would silently be a per-worker copy
        self.cache_dict = cache_dict if isinstance(cache_dict, DictProxy) else None
        if self.cache_dict is not None:
            logger.info("%s: region arrays are cached in a multiprocessing dict", type(self).__name__)
        for _, attr, subdir, _, _ in self.SOURCES:
            if getattr(config, attr, None) is None:
                setattr(config, attr, self.join_default_bu_dir(subdir))
        assert config.seq_per_img > 0, "`self.config.seq_per_img` should be greater than 0"

    def join_default_bu_dir(self, dirname):
        return os.path.join(self.config.dataset_dir, "bu", dirname)

    def _fetch(self, path, loader):
        """Load through the shared cache; new entries are only added while more than max(20 %, cache_min_free_ram) of the
        host memory is free (the reference's guard, collate.py:93-107)."""
        cache = self.cache_dict
        if cache is None:
            return loader(path)
        if path in cache:
            return cache[path]
        arr = loader(path)
        try:
            import psutil
            vm = psutil.virtual_memory()
            headroom = vm.available / vm.total
        except Exception:   # pragma: no cover
            headroom = 1.0
        if headroom > max(0.2, getattr(self.config, "cache_min_free_ram", 0.2)):
            cache[path] = arr
        return arr

    def __call__(self, batch):
        cfg = self.config
        image_paths, image_ids, _captions, all_captions, all_gts = zip(*batch)
        out = {}
        for key, attr, _, loader, with_mask in self.SOURCES:
            folder = getattr(cfg, attr)
            arrays = [self._fetch(os.path.join(folder, f"{i}.npy"), loader) for i in image_ids]
            if with_mask:
                out[key], out["att_masks"] = pad_rows(arrays, want_mask=True)
                # (not in the reference's dict) the longest region list: lets the model clip the padded regions
                # (relation_transformer.py:398-405) without reading the mask back from the device
                out["att_max_len"] = max(int(a.shape[0]) for a in arrays)
            else:
                out[key] = pad_rows(arrays)
        token_rows = []
        for caps in all_captions:                      # item order, then sample order: the reference's comprehension
            for text in random.sample(caps, min(cfg.seq_per_img, len(caps))):
                ids = self.tokenizer.encode(text, add_bos_eos=True, max_seq_length=cfg.max_seq_length)
                token_rows.append(np.asarray(ids, dtype=np.int64))
        out["seqs"], out["masks"] = pad_seqs(token_rows, 0)
        # (not in the reference's dict) decoder positions of every caption row that carry a target — BOS and the tokens, each
        # predicting its successor: len(ids) - 1 — for the valid-position decoder of NativeTrainer (ortk_batch.cap_off)
        out["cap_len"] = torch.tensor([max(1, len(r) - 1) for r in token_rows], dtype=torch.int64)
        out["gts"], out["image_paths"], out["image_ids"] = all_gts, image_paths, image_ids
        return out

    @classmethod
    def add_argparse_args(cls, parser):
        """The reference's data flags (names and defaults are the contract: collate.py:171-188, 218-227)."""
        parser.add_argument("--max_seq_length", type=int, default=18, help="caption length limit, <BOS> and <EOS> included")
        parser.add_argument("--seq_per_img", type=int, default=5, help="captions drawn per image and training step")
        for _, attr, subdir, _, _ in cls.SOURCES:
            parser.add_argument("--" + attr, type=str, default=None,
                                help=f"directory of the per-image <image_id>.npy arrays (default: <dataset_dir>/bu/{subdir})")


class UpDownCollate(_RegionBatcher):
    """Bottom-up region features only (reference ``UpDownCollate``, collate.py:77-188)."""
    SOURCES = (("att_feats", "input_att_dir", "cocobu_att", _load_features, True),)


class ObjectRelationCollate(_RegionBatcher):
    """Region features + relative boxes (reference ``ObjectRelationCollate``, collate.py:191-227)."""
    SOURCES = UpDownCollate.SOURCES + (("boxes", "input_rel_box_dir", "cocobu_box_relative", _load_boxes, False),)
