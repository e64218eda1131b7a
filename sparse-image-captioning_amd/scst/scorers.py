"""SCST reward scorer — mirror of ``sparse_caption/scst/scorers.py:17-114`` (``CaptionScorer``) on top of the native,
multi-threaded host scorer of ``libortk.so`` (``include/ortk_scorer.h``): CIDEr-D (``ciderD_scorer.py``) and per-sentence
BLEU-1..4 (``bleu_scorer.py``).  Same call contract::

    scorer = CaptionScorer(path_to_cached_tokens, cider_weight=1.0, bleu_weight=None)
    sc_sample, sc_baseline = scorer(refs, sample, baseline)      # lists of lists of caption strings

``path_to_cached_tokens`` is the reference's df pickle name (``data/<name>.p`` holding ``document_frequency`` /
``ref_len``), a path to such a pickle, ``"corpus"`` / ``None`` for on-the-fly document frequencies, or a dict
``{"document_frequency": {...}, "ref_len": n}``.  :meth:`score_ids` is the fast path for token-id tensors straight from
``model(..., mode="sample")`` (no string round trip)."""
import ctypes as C
import os
import pickle

import numpy as np

from .. import _lib as L

_P, _I32, _I64, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_SIG = {
    "ortk_scorer_create": (_P, [_I32, _D]),
    "ortk_scorer_destroy": (None, [_P]),
    "ortk_scorer_set_df": (_I32, [_P, _P, _P, _P, _I64, _D]),
    "ortk_scorer_score": (_I32, [_P, _P, _P, _I64, _P, _P, _P, _I64, _P, _P, _I32]),
}


def _lib():
    lib = L.lib()
    if not getattr(lib, "_scorer_bound", False):
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib._scorer_bound = True
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _Vocab(dict):
    """word -> id interning (ids < 65535: an n-gram packs exactly into 64 bits on the native side)."""

    def ids(self, sentence):
        out = []
        for w in sentence.split():
            i = self.get(w)
            if i is None:
                i = self[w] = len(self)
                if i >= 65534:
                    raise ValueError("more than 65534 distinct words in one scorer")
            out.append(i)
        return out


class NativeScorer:
    """Thin handle on ``ortk_scorer``: captions are lists of int ids."""

    def __init__(self, n=4, sigma=6.0, nthreads=0):
        self._lib = _lib()
        self._h = self._lib.ortk_scorer_create(n, float(sigma))
        if not self._h:
            raise ValueError("bad scorer parameters")
        self.nthreads = int(nthreads)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ortk_scorer_destroy(self._h)
            self._h = None

    def set_df(self, ngrams, counts, ref_len):
        tok = np.asarray([t for g in ngrams for t in g], dtype=np.int32)
        off = np.zeros(len(ngrams) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(g) for g in ngrams])
        df = np.asarray(counts, dtype=np.float64)
        if self._lib.ortk_scorer_set_df(self._h, _ptr(tok), _ptr(off), _ptr(df), len(ngrams), float(ref_len)) != 0:
            raise ValueError("ortk_scorer_set_df: bad n-gram table")

    def score(self, captions, hyp_cap, item_refs, cider=True, bleu=False):
        """captions: list of id lists; hyp_cap[i] = caption index of item i's hypothesis; item_refs[i] = list of caption
        indices of its references.  Returns (cider (n,) or None, bleu (4,n) or None)."""
        n = len(hyp_cap)
        tok = np.asarray([t for c in captions for t in c], dtype=np.int32)
        off = np.zeros(len(captions) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(c) for c in captions])
        hyp = np.asarray(hyp_cap, dtype=np.int64)
        ref = np.asarray([r for rs in item_refs for r in rs], dtype=np.int64)
        roff = np.zeros(n + 1, dtype=np.int64)
        roff[1:] = np.cumsum([len(rs) for rs in item_refs])
        c_out = np.zeros(n, dtype=np.float64) if cider else None
        b_out = np.zeros((4, n), dtype=np.float64) if bleu else None
        rc = self._lib.ortk_scorer_score(self._h, _ptr(tok), _ptr(off), len(captions), _ptr(hyp), _ptr(ref), _ptr(roff), n,
                                         _ptr(c_out) if cider else None, _ptr(b_out) if bleu else None, self.nthreads)
        if rc != 0:
            raise ValueError("ortk_scorer_score: bad arguments (empty reference list or token id >= 65535)")
        return c_out, b_out


class CaptionScorer:
    """``CaptionScorer`` of scst/scorers.py:17-107."""

    def __init__(self, path_to_cached_tokens, cider_weight=1.0, bleu_weight=None, nthreads=0):
        assert isinstance(cider_weight, float)
        if bleu_weight is None:
            bleu_weight = [0.0] * 4
        else:
            assert isinstance(bleu_weight, (list, tuple))
        assert len(bleu_weight) == 4
        self.path_to_cached_tokens = path_to_cached_tokens
        self.weights = {"ciderD": cider_weight, "bleu": list(bleu_weight)}
        self.vocab = _Vocab()
        self.native = NativeScorer(4, 6.0, nthreads)
        self._df_loaded = False
        self._df_space = None        # "words": the df table is keyed by word n-grams (the reference's pickles); "ids": by token ids

    # ---- document frequencies (ciderD_scorer.py:82-88)
    def _load_df(self):
        if self._df_loaded:
            return
        self._df_loaded = True
        src = self.path_to_cached_tokens
        if src is None or src == "corpus":
            return
        if isinstance(src, dict):
            table = src
        else:
            path = src if os.path.isfile(src) else os.path.join("data", src + ".p")
            with open(path, "rb") as f:
                table = pickle.load(f, encoding="latin1")
        grams, counts = [], []
        kinds = set()
        for ngram, cnt in table["document_frequency"].items():
            if all(isinstance(t, (int, np.integer)) for t in ngram):     # a table cooked in token-id space: ids are used as they are
                kinds.add("ids")
                if any(int(t) < 0 or int(t) >= 65534 for t in ngram):
                    raise ValueError("token ids of a document-frequency table must be < 65534")
                grams.append([int(t) for t in ngram])
            else:                                                        # the reference's pickles: whitespace words
                kinds.add("words")
                grams.append(self.vocab.ids(" ".join(ngram)))
            counts.append(float(cnt))
        if len(kinds) > 1:
            raise ValueError("document-frequency table mixes word and token-id n-grams")
        self._df_space = kinds.pop() if kinds else None
        self.native.set_df(grams, counts, float(table["ref_len"]))

    @staticmethod
    def input_check(inputs, same_sub_len=True):
        assert isinstance(inputs, (list, tuple))
        assert all(isinstance(_, (list, tuple)) for _ in inputs)
        if same_sub_len:
            lens = set(len(_) for _ in inputs)
            assert len(lens) == 1, f"Each image should have the same number of captions. Received captions per image: {lens}"

    def __call__(self, refs, sample, baseline=None):
        self.input_check(refs, same_sub_len=False)
        self.input_check(sample)
        assert len(refs) == len(sample), f"`ref` and `sample` have different lengths: refs = {len(refs)}, sample = {len(sample)}"
        if baseline:
            self.input_check(baseline)
            assert len(sample) == len(baseline), \
                f"`sample` and `baseline` have different lengths: sample = {len(sample)}, baseline = {len(baseline)}"
        else:
            assert baseline is None, "`baseline` should be one of: None, list or tuple."
        ids = self.vocab.ids
        ref_ids = [[ids(c) for c in r] for r in refs]
        samp_ids = [[ids(c) for c in s] for s in sample]
        base_ids = [[ids(c) for c in b] for b in baseline] if baseline else None
        return self.score_ids(ref_ids, samp_ids, base_ids)

    def score_ids(self, refs, sample, baseline=None):
        """Same as ``__call__`` on lists of token-id lists: refs[i] = reference captions of image i, sample[i] = its
        sampled captions, baseline[i] = [greedy caption].  Returns ``(sc_sample (N*ns,), sc_baseline (N*ns,))``."""
        self._load_df()
        num_baseline = len(baseline) if baseline else 0
        ns = len(sample[0])
        captions, hyp_cap, item_refs = [], [], []
        ref_idx = []
        for r in refs:                       # every reference caption is cooked once, shared by the image's items
            ref_idx.append(list(range(len(captions), len(captions) + len(r))))
            captions.extend(r)
        for i in range(num_baseline):
            assert len(baseline[i]) == 1
            hyp_cap.append(len(captions)); captions.append(baseline[i][0]); item_refs.append(ref_idx[i])
        for i in range(len(sample)):
            for j in range(ns):
                hyp_cap.append(len(captions)); captions.append(sample[i][j]); item_refs.append(ref_idx[i])
        wc, wb = self.weights["ciderD"], self.weights["bleu"]
        use_c, use_b = wc > 0, max(wb) > 0
        cider, bleu = self.native.score(captions, hyp_cap, item_refs, cider=use_c, bleu=use_b)
        scores = np.zeros(len(hyp_cap), dtype=np.float64)
        if use_c:
            scores = scores + cider * wc
        if use_b:
            for k, w in enumerate(wb):
                scores = scores + bleu[k] * w
        sc_sample = scores[num_baseline:]
        if baseline:
            sc_baseline = np.repeat(scores[:num_baseline], ns)
        else:
            sums = sc_sample.reshape([-1, ns]).sum(-1)
            sc_baseline = (np.repeat(sums, ns) - sc_sample) / (ns - 1)
        return sc_sample, sc_baseline

    def score_sequences(self, refs, sample_seq, greedy_seq=None, eos_idx=3, pad_idx=0, decode=None):
        """Token tensors straight from the model: ``sample_seq`` (N, ns, L) and ``greedy_seq`` (N, 1, L) int arrays.

        The n-grams must live in the SAME space as the document-frequency table.  The reference scores decoded strings
        (utils/training.py:239-250: ``tokenizer.decode`` then whitespace words), and its df pickles are keyed by words, so:

        * ``decode`` given (a callable ``ids -> sentence``, e.g. ``tokenizer.decode``): every row is decoded and scored
          through the string path — exactly the reference's flow; ``refs[i]`` = list of reference STRINGS (``gts``);
        * no ``decode``: rows are cut at the first EOS / PAD and scored as raw token ids; ``refs[i]`` = list of id lists.
          Only valid when the document frequencies are in token-id space too — ``"corpus"`` mode or a table with integer
          n-grams; with a word-keyed table this raises instead of silently looking up unrelated n-grams."""
        s = np.asarray(sample_seq.cpu() if hasattr(sample_seq, "cpu") else sample_seq)
        g = None if greedy_seq is None else np.asarray(greedy_seq.cpu() if hasattr(greedy_seq, "cpu") else greedy_seq)
        if decode is not None:
            sample = [[decode(r) for r in img] for img in s]
            base = None if g is None else [[decode(img[0])] for img in g]
            return self(refs, sample, base)
        self._load_df()
        if self._df_space == "words":
            raise ValueError("score_sequences without `decode`: the document-frequency table is keyed by words, token ids would "
                             "look up unrelated n-grams; pass decode=tokenizer.decode (the reference's flow), use a table cooked "
                             "in token-id space, or 'corpus' document frequencies")

        def cut(row):
            out = []
            for t in row:
                t = int(t)
                if t == eos_idx or t == pad_idx:
                    break
                out.append(t)
            return out
        sample = [[cut(r) for r in img] for img in s]
        base = None if g is None else [[cut(img[0])] for img in g]
        return self.score_ids(refs, sample, base)
