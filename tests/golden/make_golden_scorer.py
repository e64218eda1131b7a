#!/usr/bin/env python3
"""Golden fixture G6: the reference's SCST reward scorer (scst/scorers.py CaptionScorer = CIDEr-D + BLEU) run on
synthetic captions, in "corpus" mode and with a cached document-frequency table.
    python tests/golden/make_golden_scorer.py      # writes tests/golden/g6_scst_scorer.json
"""
import json
import os
import pickle
import sys
import tempfile
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ORT_REFERENCE", "/root/reference")


def sentence(rs, words, lo, hi):
    n = rs.randint(lo, hi + 1)
    return " ".join(words[i] for i in rs.randint(0, len(words), size=n))


def main():
    sys.path.insert(0, REF)
    from sparse_caption.scst.scorers import CaptionScorer
    from sparse_caption.scst.cider.pyciderevalcap.ciderD.ciderD_scorer import precook
    rs = np.random.RandomState(2024)
    words = [f"w{i}" for i in range(23)]            # small vocabulary: plenty of shared n-grams
    N, ns = 7, 3
    refs = [[sentence(rs, words, 5, 12) for _ in range(rs.randint(2, 6))] for _ in range(N)]
    sample = [[sentence(rs, words, 0 if (i == 2 and j == 1) else 3, 14) for j in range(ns)] for i in range(N)]
    sample[0][0] = refs[0][0]                        # an exact match
    baseline = [[sentence(rs, words, 4, 11)] for _ in range(N)]
    # cached document frequencies over a synthetic 40-image corpus (format of coco-train-words.p)
    corpus = [[sentence(rs, words, 5, 12) for _ in range(5)] for _ in range(40)]
    df = defaultdict(float)
    for img in corpus:
        for ng in set(ng for c in img for ng in precook(c).keys()):
            df[ng] += 1
    out = {"refs": refs, "sample": sample, "baseline": baseline, "ref_len": len(corpus),
           "df": [[list(k), v] for k, v in df.items()], "cases": []}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "data"))
        pickle.dump({"document_frequency": df, "ref_len": len(corpus)}, open(os.path.join(d, "data", "synthetic-words.p"), "wb"))
        os.chdir(d)
        try:
            # "corpus" mode is unreachable through CaptionScorer / CiderD in the reference (CiderD.compute_score calls
            # copy_empty(), which reads an attribute corpus mode never sets): drive CiderScorer directly for it
            from sparse_caption.scst.cider.pyciderevalcap.ciderD.ciderD_scorer import CiderScorer
            cs = CiderScorer(df_mode="corpus")
            for i in range(N):
                cs += (baseline[i][0], refs[i])
            for i in range(N):
                for j in range(ns):
                    cs += (sample[i][j], refs[i])
            _, arr = cs.compute_score()
            out["corpus_cider_items"] = [float(x) for x in arr]
            for mode in ("synthetic-words",):
                for cw, bw in ((1.0, None), (1.0, [0.0, 0.0, 0.0, 0.5]), (0.5, [0.1, 0.2, 0.3, 0.4])):
                    for use_base in (True, False):
                        sc = CaptionScorer(mode, cider_weight=cw, bleu_weight=bw)
                        s, b = sc(refs, sample, baseline if use_base else None)
                        out["cases"].append({"mode": mode, "cider_weight": cw, "bleu_weight": bw, "baseline": use_base,
                                             "sc_sample": [float(x) for x in s], "sc_baseline": [float(x) for x in b]})
        finally:
            os.chdir(cwd)
    json.dump(out, open(os.path.join(HERE, "g6_scst_scorer.json"), "w"))
    print("g6:", len(out["cases"]), "cases; first sample scores", out["cases"][0]["sc_sample"][:4])


if __name__ == "__main__":
    main()
