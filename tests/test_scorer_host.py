"""SCST reward scorer (host C++ in libortk.so) against goldens produced by the reference's own CaptionScorer /
CiderScorer / BleuScorer (tests/golden/make_golden_scorer.py).  Floating point in double: tolerance 1e-9 relative (numpy's
log / power kernels vs libm differ in the last bits)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g6():
    return json.load(open(os.path.join(HERE, "golden", "g6_scst_scorer.json")))


def _table(g6):
    return {"document_frequency": {tuple(k): v for k, v in g6["df"]}, "ref_len": g6["ref_len"]}


def test_caption_scorer_matches_reference(g6):
    from sparse_image_captioning_amd.scst import CaptionScorer
    assert len(g6["cases"]) == 6
    for case in g6["cases"]:
        sc = CaptionScorer(_table(g6), cider_weight=case["cider_weight"], bleu_weight=case["bleu_weight"], nthreads=3)
        s, b = sc(g6["refs"], g6["sample"], g6["baseline"] if case["baseline"] else None)
        np.testing.assert_allclose(s, case["sc_sample"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(b, case["sc_baseline"], rtol=1e-9, atol=1e-12)
    # an exact copy of a reference scores far above the rest; an empty hypothesis scores 0 and does not crash
    assert s[0] > 3 * np.median(s)


def test_corpus_mode_matches_cider_scorer(g6):
    from sparse_image_captioning_amd.scst import CaptionScorer
    sc = CaptionScorer("corpus", cider_weight=1.0)
    s, b = sc(g6["refs"], g6["sample"], g6["baseline"])
    ns = len(g6["sample"][0])
    want = np.array(g6["corpus_cider_items"])
    np.testing.assert_allclose(s, want[len(g6["baseline"]):], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b, np.repeat(want[:len(g6["baseline"])], ns), rtol=1e-9, atol=1e-12)


def test_token_sequences_and_threads_agree(g6):
    """score_sequences (token tensors cut at EOS / PAD) == string path; 1 thread == many threads."""
    from sparse_image_captioning_amd.scst import CaptionScorer
    sc = CaptionScorer(_table(g6), cider_weight=1.0, bleu_weight=[0.0, 0.0, 0.0, 1.0], nthreads=1)
    s1, b1 = sc(g6["refs"], g6["sample"], g6["baseline"])
    v = sc.vocab
    L = 20
    def row(sent):
        ids = [v[w] + 4 for w in sent.split()][:L - 1]          # shift past PAD/BOS/EOS ids
        return ids + [3] + [0] * (L - 1 - len(ids))
    sample = np.array([[row(c) for c in img] for img in g6["sample"]])
    greedy = np.array([[row(img[0])] for img in g6["baseline"]])
    refs = [[[v[w] + 4 for w in c.split()] for c in img] for img in g6["refs"]]
    sc2 = CaptionScorer({"document_frequency": {tuple(f"t{v[w] + 4}" for w in k): c for k, c in _table(g6)["document_frequency"].items()},
                         "ref_len": g6["ref_len"]}, cider_weight=1.0, bleu_weight=[0.0, 0.0, 0.0, 1.0], nthreads=8)
    # the second scorer interns the words "t<id>": feed it id lists mapped through its own vocabulary
    m = lambda ids: [sc2.vocab.ids(f"t{i}")[0] for i in ids]
    sc2._load_df()
    s2, b2 = sc2.score_ids([[m(c) for c in img] for img in refs],
                           [[m([t for t in r if t not in (0, 3)]) for r in img] for img in sample.tolist()],
                           [[m([t for t in img[0] if t not in (0, 3)])] for img in greedy.tolist()])
    np.testing.assert_allclose(s2, s1, rtol=1e-12)
    np.testing.assert_allclose(b2, b1, rtol=1e-12)
    with pytest.raises(ValueError):
        sc2.native.score([[1, 2]], [0], [[]])


def test_score_sequences_lives_in_the_df_table_space(g6):
    """The SCST reward path (NativeTrainer.scorer_reward_fn -> score_sequences) against golden G6 WITH a document-frequency
    table: decoded through a tokenizer-like `decode` it reproduces the reference's CaptionScorer values; a table cooked in
    token-id space gives the same numbers on raw ids; raw ids against the word-keyed table are refused (they would look
    up unrelated n-grams)."""
    from sparse_image_captioning_amd.scst import CaptionScorer
    case = next(c for c in g6["cases"] if c["baseline"] and c["cider_weight"] > 0)
    words = sorted({w for img in g6["refs"] + g6["sample"] + g6["baseline"] for c in img for w in c.split()} |
                   {w for k, _ in g6["df"] for w in k})
    # a tokenizer whose id order is unrelated to the pickle order: ids 4.. in reverse alphabetical order
    w2i = {w: 4 + i for i, w in enumerate(reversed(words))}
    i2w = {i: w for w, i in w2i.items()}
    L = 24
    enc = lambda sent: ([w2i[w] for w in sent.split()][:L - 1] + [3] + [0] * L)[:L]
    decode = lambda row: " ".join(i2w[int(t)] for t in list(row)[:list(row).index(3)] if int(t) > 3)
    sample = np.array([[enc(c) for c in img] for img in g6["sample"]])
    greedy = np.array([[enc(img[0])] for img in g6["baseline"]])
    assert all(len(c.split()) < L for img in g6["sample"] + g6["baseline"] for c in img)
    sc = CaptionScorer(_table(g6), cider_weight=case["cider_weight"], bleu_weight=case["bleu_weight"])
    s, b = sc.score_sequences(g6["refs"], sample, greedy, decode=decode)
    np.testing.assert_allclose(s, case["sc_sample"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b, case["sc_baseline"], rtol=1e-9, atol=1e-12)
    # raw ids against the word-keyed table: refused
    ref_ids = [[[w2i[w] for w in c.split()] for c in img] for img in g6["refs"]]
    with pytest.raises(ValueError):
        CaptionScorer(_table(g6), cider_weight=1.0).score_sequences(ref_ids, sample, greedy)
    # the same table cooked in token-id space: raw ids give the reference's numbers
    id_table = {"document_frequency": {tuple(w2i[w] for w in k): v for k, v in _table(g6)["document_frequency"].items()},
                "ref_len": g6["ref_len"]}
    sc_ids = CaptionScorer(id_table, cider_weight=case["cider_weight"], bleu_weight=case["bleu_weight"])
    s2, b2 = sc_ids.score_sequences(ref_ids, sample, greedy)
    np.testing.assert_allclose(s2, case["sc_sample"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b2, case["sc_baseline"], rtol=1e-9, atol=1e-12)
    # and the trainer's reward function is that difference
    from sparse_image_captioning_amd.training import NativeTrainer
    import torch
    fn = NativeTrainer.scorer_reward_fn(sc_ids, ref_ids)
    np.testing.assert_allclose(fn(torch.from_numpy(sample), torch.from_numpy(greedy)).numpy(),
                               (np.array(case["sc_sample"]) - np.array(case["sc_baseline"])).astype(np.float32), rtol=1e-5, atol=1e-6)
