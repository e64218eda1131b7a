#!/usr/bin/env python3
"""Golden fixture G12: the reference's UpDownCollate / ObjectRelationCollate (sparse_caption/data/collate.py:77-227) on the
synthetic feature files of common.make_collate_fixture with the stand-in tokenizer, `random` seeded.
    python tests/golden/make_golden_collate.py      # writes tests/golden/g12_collate.npz
"""
import os
import random
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common as C  # noqa: E402
from make_golden import import_reference  # noqa: E402


def main():
    import_reference()
    from sparse_caption.data.collate import UpDownCollate, ObjectRelationCollate
    from sparse_caption.utils.config import Config
    g = {}
    with tempfile.TemporaryDirectory() as root:
        items = C.make_collate_fixture(root)
        for tag, cls, spi, msl in (("or_spi2", ObjectRelationCollate, 2, 18), ("ud_spi5", UpDownCollate, 5, 9)):
            cfg = Config(input_att_dir=os.path.join(root, "att"), input_rel_box_dir=os.path.join(root, "box"), seq_per_img=spi,
                         max_seq_length=msl, dataset_dir=root)
            random.seed(1234)
            data = cls(cfg, C.StubTokenizer())(items)
            for k in ("att_feats", "att_masks", "seqs", "masks") + (("boxes",) if "boxes" in data else ()):
                g[f"{tag}/{k}"] = data[k].numpy()
            g[f"{tag}/image_ids"] = np.array(data["image_ids"])
            print(tag, {k: tuple(v.shape) for k, v in data.items() if hasattr(v, "shape")})
    np.savez_compressed(os.path.join(HERE, "g12_collate.npz"), **g)


if __name__ == "__main__":
    main()
