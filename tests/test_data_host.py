"""Host-side batch assembly (sparse_image_captioning_amd.data.collate + ortk_pad_rows / ortk_pad_seqs in libortk.so) against
the reference's UpDownCollate / ObjectRelationCollate (golden G12, tests/golden/make_golden_collate.py): same files, same
stand-in tokenizer, same `random` seed -> identical tensors.  CPU only."""
import os
import random
import re

import numpy as np
import pytest
import torch

import common as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_header_symbols_are_exported():
    import sparse_image_captioning_amd as P
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ortk_data.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(ortk_[a-z0-9_]+)\s*\(", src)))
    assert names == ["ortk_pad_rows", "ortk_pad_seqs"]
    lib = P._lib.lib()
    for n in names:
        assert hasattr(lib, n)


@pytest.mark.parametrize("tag,cls,spi,msl", [("or_spi2", "ObjectRelationCollate", 2, 18), ("ud_spi5", "UpDownCollate", 5, 9)])
def test_collate_matches_reference_golden(golden, tmp_path, tag, cls, spi, msl):
    from sparse_image_captioning_amd.data import collate as D
    from sparse_image_captioning_amd.utils.config import Config
    g = golden("g12_collate")
    root = str(tmp_path)
    items = C.make_collate_fixture(root)
    cfg = Config(input_att_dir=os.path.join(root, "att"), input_rel_box_dir=os.path.join(root, "box"), seq_per_img=spi,
                 max_seq_length=msl, dataset_dir=root)
    random.seed(1234)
    data = getattr(D, cls)(cfg, C.StubTokenizer())(items)
    keys = ("att_feats", "att_masks", "seqs", "masks") + (("boxes",) if cls == "ObjectRelationCollate" else ())
    assert ("boxes" in data) == (cls == "ObjectRelationCollate")
    for k in keys:
        ref = g[f"{tag}/{k}"]
        assert tuple(data[k].shape) == ref.shape and data[k].numpy().dtype == ref.dtype, k
        np.testing.assert_array_equal(data[k].numpy(), ref, err_msg=k)
    assert list(data["image_ids"]) == list(g[f"{tag}/image_ids"])
    assert data["gts"] == tuple(it[4] for it in items) and data["image_paths"] == tuple(it[0] for it in items)
    # default directories (collate.py:85-86,194-195)
    cfg2 = Config(input_att_dir=None, input_rel_box_dir=None, seq_per_img=1, max_seq_length=18, dataset_dir="/ds")
    D.ObjectRelationCollate(cfg2, C.StubTokenizer())
    assert cfg2.input_att_dir == "/ds/bu/cocobu_att" and cfg2.input_rel_box_dir == "/ds/bu/cocobu_box_relative"
    # through a DataLoader, as the reference's training module wires it
    random.seed(1234)
    dl = torch.utils.data.DataLoader(D.ListDataset(items), batch_size=len(items), shuffle=False, collate_fn=getattr(D, cls)(cfg, C.StubTokenizer()))
    np.testing.assert_array_equal(next(iter(dl))["att_feats"].numpy(), g[f"{tag}/att_feats"])


def test_pad_rows_edge_cases():
    from sparse_image_captioning_amd.data.collate import pad_rows, pad_seqs
    rs = np.random.RandomState(0)
    arrs = [rs.standard_normal((n, 7)).astype(np.float32) for n in (100, 10, 0, 36, 1)]     # 10-100 regions, and an empty image
    out, mask = pad_rows(arrs, want_mask=True, nthreads=3)
    ref = torch.nn.utils.rnn.pad_sequence([torch.from_numpy(a) for a in arrs], batch_first=True, padding_value=0.0)
    assert torch.equal(out, ref)
    assert torch.equal(mask, torch.nn.utils.rnn.pad_sequence([torch.ones(a.shape[0]) for a in arrs], batch_first=True))
    out1 = pad_rows(arrs[:1], nthreads=1)
    assert torch.equal(out1[0], torch.from_numpy(arrs[0]))
    seqs, m = pad_seqs([[2, 5, 3], [2, 3], [2, 9, 9, 9, 3]], pad=0)
    assert seqs.tolist() == [[2, 5, 3, 0, 0], [2, 3, 0, 0, 0], [2, 9, 9, 9, 3]] and m.sum().item() == 10
    with pytest.raises(AssertionError):
        pad_rows([np.zeros((2, 3), np.float32), np.zeros((2, 4), np.float32)])


def test_collate_fn_of_every_registered_model_is_constructible_like_the_caller_does():
    """utils/training.py:78-81 builds the loaders with ``model_cls.COLLATE_FN(config=..., tokenizer=..., cache_dict=...)``;
    ``register_into`` puts the classes (with that attribute) into the reference's registry."""
    import types
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.data.collate import ObjectRelationCollate, UpDownCollate
    cfg = types.SimpleNamespace(dataset_dir="/data", input_att_dir=None, input_rel_box_dir=None, seq_per_img=5, max_seq_length=18)
    want = {"relation_transformer": ObjectRelationCollate, "relation_transformer_prune": ObjectRelationCollate, "transformer": UpDownCollate}
    for name, cls in want.items():
        model_cls = P.get_model(name)
        assert model_cls.COLLATE_FN is cls
        c = model_cls.COLLATE_FN(config=cfg, tokenizer=object(), cache_dict=None)
        assert callable(c) and cfg.input_att_dir == "/data/bu/cocobu_att"
    assert cfg.input_rel_box_dir == "/data/bu/cocobu_box_relative"
    ref = types.SimpleNamespace(MODEL_REGISTRY={"relation_transformer": "theirs"})
    P.models.register_into(ref)
    assert ref.MODEL_REGISTRY["relation_transformer"] == "theirs"
    assert ref.MODEL_REGISTRY["relation_transformer_hip"].COLLATE_FN is ObjectRelationCollate
    assert ref.MODEL_REGISTRY["transformer_hip"].COLLATE_FN is UpDownCollate
    import argparse
    ap = argparse.ArgumentParser()
    P.get_model("relation_transformer").add_argparse_args(ap)
    ns = ap.parse_args([])
    assert ns.seq_per_img == 5 and ns.max_seq_length == 18 and ns.input_rel_box_dir is None and ns.d_model == 512
    with __import__("pytest").raises(ValueError):
        P.get_model("nope")
