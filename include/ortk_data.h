/* ortk_data.h — C-ABI of the host-side batch assembly that feeds the path (HOST code, multi-threaded; no GPU involved).
 *
 * Replaces the tensor assembly of the reference's collate functions (sparse_caption/data/collate.py):
 *   UpDownCollate.__call__           :119-169  att_feats / att_masks via torch.nn.utils.rnn.pad_sequence(batch_first, 0)
 *   ObjectRelationCollate.__call__   :202-216  boxes, padded the same way
 * i.e. a list of ragged (n_i, F) float32 arrays (10-100 detected regions per image) -> one zero-padded (B, max_i n_i, F)
 * array plus the (B, max n) validity mask.  File reading, caching, caption sampling and tokenisation stay in Python
 * (sparse-image-captioning_amd/data/collate.py); this is the part that moves 75 MB per 256-image batch.
 */
#ifndef ORTK_DATA_H
#define ORTK_DATA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* out (B, smax, F) float32 <- rows[i] (n_rows[i], F) float32, zero-padded; mask (B, smax) float32 (1 valid / 0 pad) or NULL.
 * smax >= max_i n_rows[i].  nthreads <= 0: hardware concurrency.  Returns 0, -1 on a bad argument. */
int ortk_pad_rows(const float* const* rows, const int64_t* n_rows, int64_t B, int64_t F, int64_t smax, float* out, float* mask,
                  int32_t nthreads);
/* same for int64 sequences (token ids): out (B, smax) <- seqs[i] (len[i]), pad value `pad`; mask optional */
int ortk_pad_seqs(const int64_t* const* seqs, const int64_t* len, int64_t B, int64_t smax, int64_t pad, int64_t* out, float* mask);

#ifdef __cplusplus
}
#endif
#endif /* ORTK_DATA_H */
