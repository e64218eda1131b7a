/* ortk_scorer.h — C-ABI of the SCST reward scorer (HOST code, multi-threaded; no GPU involved).
 *
 * Replaces, on integer token sequences, the pure-Python reward computation that follows sampling in the reference's
 * self-critical step (sparse_caption/utils/training.py:239-252):
 *   CaptionScorer.__call__                         scst/scorers.py:47-107
 *   CiderD / CiderScorer (CIDEr-D, n = 4, sigma 6)  scst/cider/pyciderevalcap/ciderD/ciderD_scorer.py:18-226
 *   Bleu / BleuScorer per-sentence BLEU-1..4       coco_caption/pycocoevalcap/bleu/bleu_scorer.py:24-261 ("closest" length)
 * Words are interned to int32 ids by the caller (sparse-image-captioning_amd/scst/scorers.py does it for strings), so an
 * n-gram is up to 4 ids < 65535 packed exactly into 64 bits: no hashing collisions, results equal the reference's.
 * All arithmetic is double precision in the reference's operation order.
 */
#ifndef ORTK_SCORER_H
#define ORTK_SCORER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ortk_scorer ortk_scorer;

/* n = n-gram order of CIDEr-D (the reference uses 4; 1..4 supported), sigma = Gaussian length-penalty width (6.0). */
ortk_scorer* ortk_scorer_create(int32_t n, double sigma);
void ortk_scorer_destroy(ortk_scorer* s);

/* Cached document frequencies (the reference's `coco-train-words.p`: {"document_frequency": {ngram: count}, "ref_len": D}).
 * n-gram i = tokens[key_off[i] .. key_off[i+1]), 1..4 ids each; ref_len = D (the log is taken inside, ciderD_scorer.py:88).
 * Without this call the scorer runs in "corpus" mode: document frequencies and D come from the references of each
 * ortk_scorer_score call (ciderD_scorer.py:117-128,190-192,221-226).  Returns 0, or -1 on a bad argument. */
int ortk_scorer_set_df(ortk_scorer* s, const int32_t* tokens, const int64_t* key_off, const double* df, int64_t nkeys,
                       double ref_len);

/* Scores `nitems` (hypothesis, references) items.
 *   caption c          = cap_tok[cap_off[c] .. cap_off[c+1])                 (table of ncaps captions, hypotheses and references)
 *   hypothesis of item i = caption hyp_cap[i]
 *   references of item i = captions ref_cap[item_ref_off[i] .. item_ref_off[i+1])   (>= 1 each)
 * cider_out (nitems) and/or bleu_out (4 x nitems, row k = BLEU-(k+1)) may be NULL.  nthreads <= 0: hardware concurrency.
 * Returns 0, -1 on a bad argument (a token >= 65535, an item without references, ...). */
int ortk_scorer_score(const ortk_scorer* s, const int32_t* cap_tok, const int64_t* cap_off, int64_t ncaps,
                      const int64_t* hyp_cap, const int64_t* ref_cap, const int64_t* item_ref_off, int64_t nitems,
                      double* cider_out, double* bleu_out, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif /* ORTK_SCORER_H */
