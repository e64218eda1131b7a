/*
 * ortk.h — C-ABI of libortk.so: the MI355X (gfx950) HIP implementation of the Object Relation
 * Transformer hot path of jiahuei/sparse-image-captioning.
 *
 * Boundary rules:
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*; no torch / C++ types;
 *   - caller owns every buffer (outputs and workspace are caller-allocated; sizes via *_bytes queries);
 *   - no hidden allocation, no host synchronisation, no global mutable state: every call only enqueues
 *     kernels on `stream` (thread-safe per stream, hipGraph-capturable);
 *   - return 0 on success, a negative ORTK_E* code on bad arguments, a positive hipError_t if a launch failed.
 *
 * The reference is 100 % Python/PyTorch, so there is no native FFI to mirror; each entry point cites the
 * reference Python function whose arithmetic it replaces (paths relative to the reference repo root).
 * All floating-point tensors are fp32 in memory.  `precision` selects the MFMA operand type of the dense
 * projections only: 0 = fp32 MFMA (bit-faithful parity mode), 1 = bf16 MFMA with fp32 accumulation.
 */
#ifndef ORTK_H
#define ORTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORTK_VERSION 2      /* bumped whenever a struct or a signature of this header changes: a binding must refuse another version */
#define ORTK_EINVAL (-1)   /* bad argument / unsupported shape */
#define ORTK_ENOSPC (-2)   /* workspace too small */
#define ORTK_ENOSYS (-3)   /* option not implemented (e.g. ACORT weight sharing) */
#define ORTK_EEXCHANGE (-4) /* ortk_decode_status: an exchange group of the column-split stack kernel never met (launch not fully resident) */

typedef void* ortk_stream;  /* hipStream_t */

int ortk_version(void);
/* 1 if a gfx950 device is visible, else 0 (never throws). */
int ortk_device_ok(void);

/* ------------------------------------------------------------------------------------------------
 * Model geometry — the fields models/transformer.py:418-437 reads from `config`.
 * ---------------------------------------------------------------------------------------------- */
struct ortk_sparse_plan;
typedef struct ortk_config {
    int32_t d_model, d_ff, n_layers, n_heads;
    int32_t vocab, feat, seq_len;              /* vocab_size, att_feat_size, max_seq_length          */
    int32_t pad_id, bos_id, eos_id, unk_id;
    int32_t box_trig;                          /* !no_box_trigonometric_embedding: 1 = 64-d sin/cos embedding, 0 = 4-d log-ratios */
    int32_t precision;                         /* 0 fp32 MFMA, 1 bf16 MFMA                           */
    float   drop_src, drop;                    /* att_embed dropout; every other dropout (0.1 | 0.1/3)*/
    /* ACORT layer sharing (`share_layer_encoder / share_layer_decoder`, relation_transformer.py:80-88, transformer.py:175-183):
     * 0 = layer l has its own weights; k > 0 = layer l IS layer k-1 (k-1 < l, itself unshared): same arena offsets, the
     * gradients of both positions accumulate there, the arena holds one entry set per distinct layer. */
    int32_t share_enc[16], share_dec[16];
    /* ACORT projection sharing inside an attention module (`share_att_encoder / share_att_decoder`,
     * relation_transformer.py:140-142,162-175; transformer.py:223-225,258-263): 0 = W_q, W_k, W_v, W_o (linears.0-3);
     * 1 = "kv": K = V = linears.1(key), W_o = linears.2;  2 = "qk": Q = linears.0(query), K = linears.0(key),
     * V = linears.1(value), W_o = linears.2.  Three linears per module in the state_dict, as in the reference. */
    int32_t share_att_enc, share_att_dec;
    /* 1 = the plain `transformer` model (models/transformer.py:617-665) instead of the relation transformer: encoder self-
     * attention without geometry bias (no WGs parameters; `boxes` is ignored and may be NULL), region embedding without the
     * zeroing of padded regions (src_embed = Linear + ReLU + Dropout on every row), and the reference's state_dict names
     * for that class (`core.src_embed.0.*`, `core.encoder.*`, `core.decoder.*`, `core.tgt_embed.*`, `core.generator.proj.*`). */
    int32_t no_box;
    /* Execution option, not geometry (may be NULL): sparse plans over blocks of the weight arena for the TRAINING step
     * (`relation_transformer_prune`: the masked linears of pruning/masked_layer.py:84-110,134-135 as sparse products).
     * sparse_fwd: blocks = (N,K) weight blocks at their arena offsets; ortk_forward rebuilds it from the effective weights
     * of the call (s*W: the mask sample of THIS step) and runs those projections with ortk_spmm.  sparse_bwd (mixed
     * precision only): the same blocks TRANSPOSED ((K,N), same offsets, ld = N): rebuilt by ortk_forward from its
     * transposed bf16 weight copies, used by ortk_backward for the data gradients dX = dY W~.  The weight gradients stay
     * dense MFMA products: the straight-through mask gradient needs dLoss/d(s*W) at EVERY position (sampler.py:10-34). */
    const struct ortk_sparse_plan* sparse_fwd;
    const struct ortk_sparse_plan* sparse_bwd;
} ortk_config;

/* ------------------------------------------------------------------------------------------------
 * Parameter arena.  All parameters live in ONE flat fp32 buffer (gradients, Adam moments and — for the
 * `_prune` variant — mask logits use mirrors with identical offsets).  Q/K/V weights of a layer are
 * adjacent (one N=3d GEMM), the cross-attention K/V weights of ALL decoder layers are adjacent (one
 * GEMM per batch of images), the 8 WG geometry vectors of a layer form one (8,64) block.
 * Entry names are the reference state_dict keys.
 * ---------------------------------------------------------------------------------------------- */
int64_t ortk_arena_numel(const ortk_config* cfg);               /* trainable floats (size of grad / Adam mirrors) */
int64_t ortk_arena_numel_with_buffers(const ortk_config* cfg);  /* + the `pe` buffer stored behind them           */
int32_t ortk_arena_entries(const ortk_config* cfg);
/* name_buf >= 128 bytes; shape[4]; kind: 0 = parameter, 1 = parameter that the `_prune` variant masks (>= 2-D),
 * 2 = buffer (model.tgt_embed.1.pe, transformer.py:365-376: filled by the host, never trained). */
int ortk_arena_entry(const ortk_config* cfg, int32_t index, char* name_buf, int64_t* offset,
                     int64_t* numel, int32_t* ndim, int64_t* shape, int32_t* kind);

/* ------------------------------------------------------------------------------------------------
 * Whole-path entry points (the native executor).
 * ---------------------------------------------------------------------------------------------- */
/* The 32-bit dropout key of one site of a training-mode ortk_forward(seed): stack 0 = the region embedding's dropout
 * (relation_transformer.py:331-333; element index row*d + col over the (B*S, d) output), 1 = the PositionalEncoding's
 * (transformer.py:398-401; (R*T, d)), 2 / 3 = encoder / decoder layer `layer` with k = 0 self-attention probabilities
 * ((group, head, query, key) flattened as the kernel's P tensor), 1 self-attention sublayer output, [decoder: 2 cross-attention
 * probabilities ((image, head, caption*T + t, region)), 3 cross-attention sublayer output, 4 FFN hidden, 5 FFN sublayer output;
 * encoder: 2 FFN hidden, 3 FFN sublayer output].  Element i of that site keeps iff ortk_dropout_apply(ones, .., n, p, key)[i]
 * != 0 — the hook the train-mode parity test uses to replay the SAME masks through the oracle. */
uint32_t ortk_dropout_site_seed(uint64_t seed, int32_t stack, int32_t layer, int32_t k);

typedef struct ortk_batch {
    const float*   att_feats;   /* (B, S, feat)   zero padded                       collate.py:119-169 */
    const float*   boxes;       /* (B, S, 4)      relative x0,y0,x1,y1                                  */
    const float*   att_masks;   /* (B, S)         1 = valid region                                      */
    const int64_t* seqs;        /* (R, seq_stride) token ids; columns 0..T-1 are decoder inputs,        */
    int64_t        seq_stride;  /*                 columns 1..T the targets (T = teacher-forced steps)  */
    const float*   tok_weight;  /* (R, T) per-target weight: XE -> masks[:,1:]; SCST -> mask*reward     */
    int32_t B, S, R, T;         /* R = B * captions-per-image                                           */
    /* Optional (all three or none; mixed precision, fused criterion only — ortk_forward with logp_out != NULL refuses them):
     * VALID-POSITION layout of the decoder rows.  The reference runs its decoder over all T positions of every caption and
     * masks the padded ones out of the loss (transformer.py:187-210, utils/losses.py:36-43); a padded position never feeds a
     * valid one (causal self-attention), so the decoder here runs on the valid PREFIX of every caption only: caption r owns
     * rows [cap_off[r], cap_off[r+1]) of a compact row space of Mc rows (cap_off: device, R+1 entries, cap_off[0] = 0,
     * 1 <= length <= T; the captions of an image stay adjacent), row_pos[i] = r*T + t of compact row i (device, Mc entries).
     * The caller guarantees tok_weight[r][t] == 0 for every position outside the prefix.  Same loss and gradients as the
     * padded layout, in train mode too: every dropout site of compact row i draws as row row_pos[i] of the padded layout. */
    const int32_t* cap_off;
    const int32_t* row_pos;
    int32_t        Mc;
    /* 1 = the captions are ROLLOUTS (the update pass of an SCST step): the decoder's self-attention masks nothing but the
     * future.  The reference differentiates its sampled captions through the cached incremental passes that drew them, and a
     * cached step attends to EVERY earlier position (transformer.py:265-269: mask = None once a cache exists) — also to one
     * whose sampled token happens to carry the PAD id, which the teacher-forced key mask (seq != pad,
     * relation_transformer.py:356-358) would hide.  0 = that key mask (ground-truth captions: PAD only pads). */
    int32_t        no_pad_keys;
} ortk_batch;

/* The two tables of the valid-position layout from the per-caption position counts (device, int64, 1 <= counts[r] <= T; R <=
 * 65 536): cap_off (R + 1 ints; cap_off[R] = Mc = sum of the counts, which the HOST must know too — it sizes every launch) and
 * row_pos (Mc ints): row_pos[cap_off[r] + t] = r * T + t.  Two small launches; nothing but the counts has to cross PCIe. */
int ortk_valid_position_tables(const int64_t* counts, int32_t R, int32_t T, int32_t* cap_off, int32_t* row_pos, ortk_stream stream);
size_t ortk_train_workspace_bytes(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T);
/* 1 if a batch of this geometry may carry the valid-position layout (cap_off / row_pos / Mc), else 0. */
int ortk_valid_positions_ok(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T);

/* Teacher-forced forward (RelationTransformerModel._forward, models/relation_transformer.py:368-372).
 * Activations stay in `ws` for a following backward.  If logp_out != NULL the (R,T,ldv) log-probs are
 * written there (ldv >= vocab, multiple of 4).  `train` != 0 enables dropout keyed by `seed`. */
int ortk_forward(const ortk_config* cfg, const float* params, const ortk_batch* batch, void* ws, size_t ws_bytes,
                 float* logp_out, int64_t ldv, int32_t train, uint64_t seed, ortk_stream stream);
/* The same in two calls on ONE workspace: phase 1 = weight copies + encoder (batch->seqs may be NULL; R and T still shape the
 * workspace), phase 2 = decoder + generator on the encoder state phase 1 left there (same params, train, seed); phase 0 = all.
 * Between the two the encoder memory — (B*S, d_model) rows, *dtype = 0 fp32 / 1 bf16 — can feed a decode of the same images
 * (ortk_decode_opts.memory): an SCST step (utils/training.py:202-255) then runs its encoder once instead of twice. */
int ortk_forward_phase(const ortk_config* cfg, const float* params, const ortk_batch* batch, void* ws, size_t ws_bytes,
                       float* logp_out, int64_t ldv, int32_t train, uint64_t seed, int32_t phase, ortk_stream stream);
void* ortk_train_workspace_memory(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T, void* ws, int32_t* dtype);

/* Fused criterion on the logits left in `ws` by ortk_forward:
 *   loss = -sum_{r,t} logp[r,t,target]*tok_weight[r,t] / norm      (LanguageModelCriterion losses.py:36-43,
 *   RewardCriterion losses.py:15-29), norm = *norm_dev (device scalar, e.g. sum of the 0/1 mask).
 * Writes the scalar loss to *loss_dev and leaves dLoss/dlogits in `ws`. */
int ortk_loss(const ortk_config* cfg, const ortk_batch* batch, void* ws, size_t ws_bytes,
              const float* norm_dev, float* loss_dev, ortk_stream stream);

/* Alternative to ortk_loss for an external criterion: dlogp (R,T,ldv) -> dLoss/dlogits in `ws`. */
int ortk_loss_external(const ortk_config* cfg, const ortk_batch* batch, void* ws, size_t ws_bytes,
                       const float* logp, const float* dlogp, int64_t ldv, ortk_stream stream);

/* Backward through decoder, encoder, geometry bias and att_embed; ACCUMULATES into `grads` (arena mirror). */
int ortk_backward(const ortk_config* cfg, const float* params, float* grads, const ortk_batch* batch,
                  void* ws, size_t ws_bytes, int32_t train, uint64_t seed, ortk_stream stream);
/* The same in two halves, so that a data-parallel host can start the all-reduce of the decoder half's gradients while
 * the encoder half still runs: phase 1 = generator + decoder stack + token embedding + cross-attention K/V projections
 * (afterwards every gradient at arena offsets >= ortk_arena_decoder_offset(cfg) is final), phase 2 = encoder stack,
 * geometry bias, att_embed (must follow phase 1 on the same workspace); phase 0 = both (== ortk_backward). */
int ortk_backward_phase(const ortk_config* cfg, const float* params, float* grads, const ortk_batch* batch,
                        void* ws, size_t ws_bytes, int32_t train, uint64_t seed, int32_t phase, ortk_stream stream);
int64_t ortk_arena_decoder_offset(const ortk_config* cfg);

/* Cached-attention decoding: CachedTransformerBase._generate_captions (models/transformer.py:471-561)
 * + CaptionModel.batch_beam_search (models/caption_model.py:30-226, group_size 1). */
/* ------------------------------------------------------------------------------------------------
 * Sparse weights.  A *plan* names pruned weight blocks and owns the device buffers of their sparse images; the images are
 * rebuilt ON THE DEVICE from the dense (zero-filled) weights of the current call (ortk_sparse_build: no host sync), so
 * there is no host-side cache that could go stale.  A block is an (N outputs, K inputs) row-major matrix: a torch Linear
 * weight, or — for the data gradient dX = dY W — its transposed bf16 copy (then N = in_features).
 *
 * format ORTK_SP_GU16 (mixed precision; the fast path): "group union".  The outputs are cut in groups of 16 columns
 *   (one MFMA tile) and the inputs in chunks of 512.  For each (group g, chunk c) — slot = slot0 + g*nchunks + c — the image
 *   holds the UNION of the input columns in which any of the 16 outputs has a non-zero, as up to 16 k-steps of 32 columns:
 *     nsteps[slot]            number of k-steps
 *     kofs[(slot*16 + s)*64 + lane]   the two LDS rows (chunk-relative input columns, lo / hi 16 bits) whose addresses lane
 *                             `lane` supplies to the two transposing LDS reads of step s
 *     wfrag[((slot*16 + s)*64 + lane)*8 + j]  bf16 weights in MFMA operand order: output 16g + (lane & 15), union entry
 *                             8*(lane >> 4) + j of the step
 *   The product is then a DENSE bf16 MFMA over the compacted K (56 % of K at 95 % sparsity, 33 % at 97.5 %), its activation
 *   operand gathered from an LDS tile by ds_read_b64_tr_b16.  Entry e of a step takes an input column with
 *   k mod 8 == ((e & 7) + 4*((e >> 3) & 1)) & 7 (zero-weight dummies pad the residue classes): the gathers are then
 *   bank-conflict free.  Worst-case sized (16 steps per slot): nothing can overflow.
 * format ORTK_SP_ELL32 (fp32 parity mode) / ORTK_SP_ELL16: sorted, padded ELL.  The N output columns are cut in ranges of
 *   512; inside a range they are ordered by their number of non-zeros (descending) and grouped in chunks of 64: chunk c =
 *   64 output columns, one per lane, padded to chunk_len[c] entries.  ELL32: entry j of lane l of chunk c =
 *   stream[chunk_ptr[c] + j*64 + l] = uint32x2 { slot(k)*16, fp32 bits of the value } (chunk_len multiple of 4).  ELL16:
 *   entry PAIRS uint32x2 { off_even | off_odd << 16, bf16 w_even | w_odd << 16 } at stream[chunk_ptr[c]/2 + (j/2)*64 + l]
 *   (chunk_len multiple of 8).  slot(k) = (k & ~7) | ((k + (k >> 3)) & 7) is the 16-byte slot of input column k in the
 *   kernel's LDS image (K <= 2048).  perm[c*64 + l] = the output column of lane l (-1 = pad lane).  Capacity-planned: the
 *   builder drops what does not fit and raises *overflow.
 * ---------------------------------------------------------------------------------------------- */
#define ORTK_SP_ELL32 0
#define ORTK_SP_ELL16 1
#define ORTK_SP_GU16  2
typedef struct ortk_sparse_block {
    int64_t src_offset;        /* element offset of the dense (N,K) block inside the dense buffer handed to the builder   */
    int64_t ld;                /* its leading dimension (elements)                                                         */
    int64_t stream_offset;     /* ELL: first entry of this block's region in the stream buffer                            */
    int64_t capacity;          /* ELL: entries reserved for it                                                            */
    int32_t N, K;
    int32_t chunk0;            /* ELL: index of its first chunk (ceil(N/64) chunks);  GU16: slot0, its first (group, chunk) slot
                                  (ceil(N/16) * ceil(K/512) slots)                                                        */
    int32_t row0;              /* ELL: index of its first output column in the builder's count scratch                   */
} ortk_sparse_block;

/* One set of sparse blocks (device buffers owned by the caller; the block table is given twice: host copy for the
 * launch geometry, device copy for the kernels). */
typedef struct ortk_sparse_plan {
    const ortk_sparse_block* blocks_host; const ortk_sparse_block* blocks_dev;
    int32_t nblocks, format;
    void* stream;              /* ELL: entries;  GU16: wfrag (bf16, total_slots*16*64*8)                                  */
    int32_t* chunk_ptr;        /* ELL: per chunk; GU16: kofs (uint32, total_slots*16*64)                                  */
    int32_t* chunk_len;        /* ELL: per chunk; GU16: nsteps (per slot)                                                 */
    int32_t* perm;             /* ELL: per lane slot; GU16: per block, the maximum of its nsteps (written by the builder: the
                                  product kernel runs every group of a block for that many k-steps, branch-free)           */
    int32_t* count_scratch;    /* ELL: >= total_rows; GU16: non-zeros per slot (statistics)                               */
    int32_t* overflow;         /* device int: set to 1 by the ELL builder when a block exceeded its capacity              */
    int64_t total_rows;        /* ELL: sum of N over the blocks; GU16: total number of slots                              */
} ortk_sparse_plan;

/* Build every block of the plan from `dense` (dtype 0 fp32 / 1 bf16): no host sync. */
int ortk_sparse_build(const ortk_sparse_plan* plan, const void* dense, int32_t dtype, ortk_stream stream);

/* Y = epi(X W~^T) for block `block` of a built plan.  Same epilogue as ortk_gemm:
 *   v = acc + bias[n]; relu; v *= rowscale[m]; dropout(p, seed, index m*N+n); v *= (gate[m,n]>0)*gate_scale; v += resid[m,n].
 * Replaces F.linear on zero-filled pruned weights (scripts/eval_model.py:64-88) and the masked linears of
 * pruning/masked_layer.py:84-110,134-135 (forward, and the data gradient with a plan built from W^T). */
typedef struct ortk_spmm_args {
    const void* X; void* Y;
    int64_t ldx, ldy; int64_t M;
    int32_t x_dtype, y_dtype;
    const float* bias; const float* rowscale; const float* resid; int64_t ldr;
    const void* gate; int64_t ldg; int32_t gate_dtype; float gate_scale;
    int32_t relu; float drop_p; uint32_t drop_seed;
    /* optional (rows ints): output row m takes the dropout draws of row drop_rows[m] — the valid-position decoder layout
     * (ortk_batch.row_pos) draws what the padded (caption, position) layout draws, so a step is the same function of its seed in
     * both layouts and an SCST update on the valid positions reproduces the masks of its train-mode rollout. */
    const int32_t* drop_rows;
} ortk_spmm_args;
int ortk_spmm(const ortk_sparse_plan* plan, int32_t block, const ortk_spmm_args* a, ortk_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Rows-stationary chain of the row-wise operators between two attention calls (csrc/ortk_chain.hip; mixed precision, d_model 512):
 *
 *   x  = x_in                                                                 (M, 512) fp32 residual rows
 *   R  (a_in != NULL):  x += dropout(a_in W_r^T + bias_r)          -> x_mid   SublayerConnection around an attention (transformer.py:293-294)
 *   L1 (g1 != NULL):    y = LayerNorm(x; g1, b1)                   -> y1 (bf16), st1 {mean, std} per row (transformer.py:338-341)
 *   S1 (n1 <= 3):       out1[:, 512 i ..] = y W_i^T + bias_s1[512 i ..]       projections of y (packed QKV, cross-attention query)
 *   F  (NC = d_ff / 512 > 0):  h = dropout(relu(y W1^T + bias_h)) -> h (bf16, (M, d_ff));  x += dropout(h W2^T + bias_o) -> x_out
 *   L2 (g2 != NULL), S2 (n2 <= 3): as L1 / S1 on the new x            -> y2, st2, out2
 *
 * One launch, the rows stay in registers / LDS, the weights stream; every store is what the separate kernels (ortk_layernorm_fwd,
 * ortk_gemm with its bias / ReLU / dropout / residual epilogue) leave for the backward pass, and the dropout draws are the GEMM
 * epilogue's (element (row, column) of the (M, N) output under seed_r / seed_h / seed_o), so ortk_backward does not care which
 * executor ran the forward.  Weights: `n_units` 512 x 512 blocks of the bf16 arena `w16`, in stream order [R] [S1 ..] [W1 chunk c,
 * W2 chunk c] x NC [S2 ..]; unit = {offset of its first output row (elements), leading dimension}: rows = outputs, 512 inputs from
 * the offset on.  `units_dev` is a DEVICE array; `packed` a scratch of ortk_chain_packed_bytes(n_units). */
typedef struct ortk_chain_unit { int64_t offset, ld; } ortk_chain_unit;
typedef struct ortk_chain_args {
    const void* w16; const ortk_chain_unit* units_dev; int32_t n_units;
    void* packed; size_t packed_bytes;
    int64_t M;
    const float* x_in;
    const void* a_in; const float* bias_r; float* x_mid; uint32_t seed_r;
    const float *g1, *b1; void* y1; float* st1;
    int32_t n1; const float* bias_s1; void* out1; int64_t ld1;
    int32_t NC; const float *bias_h, *bias_o; void* h; float* x_out; uint32_t seed_h, seed_o;    /* h, y1, y2, st1, st2 may be NULL: not kept (inference) */
    const float *g2, *b2; void* y2; float* st2;
    int32_t n2; const float* bias_s2; void* out2; int64_t ld2;
    float drop_p, eps;
    int32_t* progress;      /* optional: 16 ints of device scratch (zeroed by the call) -> 8 L2 prefetcher workgroups pace the weight stream */
    const int32_t* drop_rows;   /* optional (M ints): row m draws its dropout (seed_r / seed_h / seed_o) as row drop_rows[m] (see ortk_gemm_args) */
} ortk_chain_args;
size_t ortk_chain_packed_bytes(int32_t n_units);
int ortk_row_chain(const ortk_chain_args* a, ortk_stream stream);

/* The backward counterpart: everything between two attention-backward calls except the weight gradients (which reduce over all
 * rows and stay GEMMs), through the TRANSPOSED bf16 weight copies `w16t` (block (N, K) of the arena stored (K, N) at the same offset):
 *
 *   P0 (nin = 1 | 3):  gy = sum_i ain[:, 512 i ..] U_i^T          data gradient of a projection (dq . Wcq; packed dQ|dK|dV . Wqkv)
 *        (nin = 0):    the operand rows are dz0 (a dropout-masked gradient that already exists)
 *   LNa (nin > 0):     dx = LayerNorm'(gy; xa, sta, ga) + dresa -> dxa;  daa / dba += parameter gradients;
 *                      mask_a: dz = dropout-mask(dx; seed_a) -> operand rows of P1 / P2 (+ dza)
 *   P1 (NC > 0):       gh = gate(hgate) * (dz W2)  -> gh (bf16, (M, d_ff));  gy2 = gh W1        PositionwiseFeedForward backward
 *   LNb (NC > 0):      as LNa on gy2 with xb, stb, gb, dresb (may be the dxa rows this call has just written) -> dxb, dab, dbb, dzb
 *   P2 (n2 = 1):       out2 = dz W (through an attention's out-projection) -> bf16 rows
 *
 * Same results as ortk_gemm (data-gradient layout, gate epilogue) + ortk_layernorm_bwd_drop; units in stream order
 * [P0 x nin] [W2^T chunk c, W1^T chunk c] x NC [P2], each a 512 x 512 block of w16t: rows = OUTPUT columns of the product. */
typedef struct ortk_bchain_args {
    const void* w16t; const ortk_chain_unit* units_dev; int32_t n_units;
    void* packed; size_t packed_bytes;
    int64_t M;
    int32_t nin; const void* ain; int64_t ld_ain; const void* dz0;
    const float *xa, *sta, *ga, *dresa; float *dxa, *daa, *dba; void* dza; uint32_t seed_a; int32_t mask_a;
    int32_t NC; const void* hgate; void* gh; float gate_scale;
    const float *xb, *stb, *gb, *dresb; float *dxb, *dab, *dbb; void* dzb; uint32_t seed_b; int32_t mask_b;
    int32_t n2; void* out2;
    float drop_p, eps;
    const int32_t* drop_rows;   /* optional (M ints): the masked copies dza / dzb / dz0 draw row m as row drop_rows[m] (see ortk_gemm_args) */
} ortk_bchain_args;
int ortk_row_bchain(const ortk_bchain_args* a, ortk_stream stream);

/* Process-wide A/B switches for measurements (the scripts under scratch/); the defaults are the product path and nothing in the library reads the
 * environment.  Not thread-safe against running calls: set them before the work starts. */
typedef struct ortk_tuning {
    int32_t gemm_impl;       /* 0 automatic | 1 register-staged GEMM kernel only | 2 128 x 128 LDS-DMA tiles for every layout | 3 256 x 256 whenever legal */
    int32_t gemm_t64;        /* 64 x 64 LDS-DMA tiles while the 128 x 128 grid has at most this many workgroups (640; -1 never) */
    int32_t attn_impl;       /* 0 automatic | 1 wave kernels | 3 fp32-MFMA kernels | 4 small register-only kernels (ortk_attn.hip dispatch) */
    int32_t attn16_min_lq;   /* fp32-input query blocks shorter than this stay off the bf16-operand attention kernels (33) */
    int32_t side_stream;     /* 1: the executor queues weight gradients and other independent work on a second stream (default) | 0 */
    int32_t row_chain;       /* rows-stationary chains (ortk_row_chain / ortk_row_bchain) in the executor: 0 none (one launch per operator) | 1 forward passes
                                (default) | 2 + the encoder's backward | 3 + the decoder's backward */
    int32_t chain_wide;      /* 1 (default): forward chains of 12 289 .. 19 456 rows run the 76-row form of the kernel — one round of workgroups
                                instead of two (profiles/r04_row_chains.txt) | 0: the 48-row form everywhere | 2: the 76-row form wherever it saves
                                a round (A/B: two instead of three at the decode's 36 864 encoder rows measures the same) */
    int32_t spmm_alias;      /* 1 (default): ELL products with more than 512 input columns and one output range per workgroup keep their output
                                tile over the staged X planes (two workgroups per compute unit instead of one) | 0 */
    int32_t f32_split;       /* fp32 products of the forward layout (precision 0: the fp32 parity mode): 1 (default) on the bf16 matrix cores, every
                                operand split into three bf16 parts and six partial products kept (fp32-level error, ortk_gemm.hip:
                                gemm_f32x3_kernel / gemm_f32x3p_kernel) | 0 the fp32 MFMA kernel | 2..7 as 1 with a fixed kernel instance
                                (single-buffered 64x64, 128x64, 128x128; pipelined 64x64, 128x64, 256x128).
                                PRECONDITION of the split forms: finite operands with |x| < 3.39e38 (bf16(x) must not round to infinity).  An
                                infinite or FLT_MAX-class operand makes the second and third parts NaN (x - inf), where the fp32 MFMA kernel
                                (f32_split = 0) propagates the infinity; the path's operands (activations, weights, gradients under
                                clip_grad_value_) are far inside the range, and a guard costs two vector instructions per staged element */
    int32_t wgrad_wgs;       /* workgroups a weight-gradient GEMM with fewer than 256 output tiles is split into along K (384: tuned with the kernel alone
                                on the chip; fewer = fewer split-K atomics, which run at 1.3 TB/s at the memory side) */
    int32_t wgrad_group;     /* grouped weight gradients (ortk_wgrad_group: the weight gradients of a layer in one launch on the side stream), bit mask:
                                1 the encoder / decoder layers | 2 the generator | 4 the memory's K|V projection | 8 the row ranges of a tile meet in
                                memory (workspace) instead of adding with atomics.  0 = one launch per projection (ortk_gemm).  Mixed precision only */
    int32_t wgrad_group_splitk;   /* row ranges of every grouped launch of the executor: 0 automatic (the two fields below) | 1..8 */
    int32_t wgrad_group_wgs;      /* workgroups a grouped launch of the executor is given (row ranges = this / its 256 x 256 tiles, rounded; 80: the
                                     launch shares the chip with the caller's stream instead of filling it, scratch/wgrad_group_ab.py) */
    int32_t wgrad_group_tail;     /* 1 (default): the last group before the caller's stream waits for the side stream gets a full round of workgroups | 0 */
    int32_t feats_bf16;           /* 1 (default, mixed precision with the side stream): a bf16 copy of the region features per step (att_embed and its weight
                                     gradient on the bf16-operand kernels; the weight gradient in the backward's last grouped launch) and the chain weights
                                     packed on a third queue beside att_embed | 0: fp32 features, everything at the head of the forward on the caller's stream */
    int32_t ln_fuse;              /* bit 0: the executor's backward runs a d_model-wide data gradient and the LayerNorm backward behind it as ONE launch
                                     (ortk_gemm ln_mode 2 on short row panels).  OFF by default: alone the fused launch ties the two it replaces
                                     (59.5 vs 62.0 us at 16 640 x 512 x 512), inside the step it needs a free compute unit per workgroup and waits for
                                     the units the side stream's weight gradients hold: 11.65 vs 10.40 ms per XE step (scratch/wgrad_group_ab.py)
                                     | bit 1: ln_mode 2 on the 128-row panels of round 3 (measurement) | bit 2: the LayerNorm backward of width 512 with four
                                     instead of eight consecutive columns per lane (measurement) | bit 3: the executor keeps the LayerNorm output gradients
                                     (data gradient -> ortk_layernorm_bwd_dt) in fp32 in mixed precision too; default: bf16 there, as every other gradient
                                     that is a GEMM operand in that mode */
    int32_t samp_epilogue;        /* 1 (default): sampling decodes in mixed precision take their tokens from the generator GEMM's epilogue (Gumbel-max candidates
                                     + soft-max partials per 64 logits, ortk_gemm_args.tile_samp) and never store the logit rows | 0: logits + sample step */
    int32_t gemm_epilogue;        /* 0 (default): the forward-layout LDS-DMA GEMM kernels run the lean epilogue (options the launcher has verified compiled
                                     out, dropout / gate as kernel instances; bf16 results of the 256 x 256 tile stored 16 bytes per lane) | bit 0: the general
                                     epilogue of rounds 2-5 everywhere (measurement: profiles/r06_gemm_epilogue.txt) | bit 1: a column count of 256 n + 128
                                     (the padded vocabulary) is NOT split into a 256 x 256-tile launch + a 128-column remainder (measurement) */
} ortk_tuning;
void ortk_get_tuning(ortk_tuning* out);
int ortk_set_tuning(const ortk_tuning* t);

#define ORTK_DEC_UNFUSED 1
#define ORTK_DEC_STACK 2
#define ORTK_DEC_SPARSE_STREAM 4
#define ORTK_DEC_STACK_RB20 8
#define ORTK_DEC_STACK_SPLIT 16
#define ORTK_DEC_SPLIT_SMALL 32
#define ORTK_DEC_SPARSE_GATHER 64
typedef struct ortk_decode_opts {
    int32_t beam_size;            /* 1 = greedy; >1 = beam search; <1 with num_random_sample > 0 = multinomial */
    int32_t num_random_sample;
    float   temperature;
    int32_t decoding_constraint;  /* forbid repeating the previous token */
    int32_t length_penalty;       /* 0 none, 1 "wu_<alpha>", 2 "avg_<alpha>"   utils/model_utils.py:121-146 */
    double  length_alpha;
    uint64_t seed;                /* multinomial: Gumbel-max over counter-based uniforms */
    /* Optional (may be NULL): sparse plan over blocks of the weight arena (src_offset = arena offset of the block, see
     * ortk_linear_block).  ortk_decode rebuilds it from the weights of this call and runs every projection that has a
     * block in it as a sparse product (ortk_spmm) instead of a dense GEMM. */
    const struct ortk_sparse_plan* sparse;
    /* Executor choice (bit flags, 0 = automatic): by default a decode of >= 1 600 rows (images x beams) in mixed precision runs
     * the one-launch-per-position decoder stack kernel on the dense weight stream, smaller ones the unfused executor.
     *   ORTK_DEC_UNFUSED        never the stack kernel;
     *   ORTK_DEC_STACK          the stack kernel whenever the configuration is served, whatever the row count;
     *   ORTK_DEC_SPARSE_STREAM  the decoder weights of this call are mostly zeros (a pruned checkpoint evaluated as dense
     *                           linears on zero-filled weights, scripts/eval_model.py:64-88): the stack kernel streams their
     *                           NON-ZEROS (rebuilt on the device from the weights of the call, no host sync; correct at any
     *                           density, faster than the dense stream above ~80 % zeros); implies ORTK_DEC_STACK;
     *   ORTK_DEC_SPARSE_GATHER  with ORTK_DEC_SPARSE_STREAM: the non-zeros as per-output-column gather lists over transposed operand
     *                           images instead of scatter entries expanded for the matrix cores — work proportional to the
     *                           non-zeros (the longest column of each group of 64); pays from ~98.5 % zeros on (the reference's published
     *                           98.8 / 99.1 % models; at 97.5 % the scatter stream is the faster one); same results up to fp32
     *                           summation order;
     *   ORTK_DEC_STACK_RB20     dense stream with 20-row workgroups (measurement);
     *   ORTK_DEC_STACK_SPLIT    the column-split form of the stack kernel: groups of 2 / 4 / 8 workgroups of one XCD share 64 rows
     *                           and split every projection's output columns (each streams 1/2 .. 1/8 of the weights; partial
     *                           results are exchanged through that XCD's L2).  Dense stream; implies ORTK_DEC_STACK; row counts
     *                           whose groups do not all fit the chip at once (> 8 192 rows) run the plain stack kernel;
     *   ORTK_DEC_SPLIT_SMALL    automatic choice, but decodes of at most 4 096 rows (8 workgroups per group up to 2 048 rows, 4
     *                           above) take the column-split form: the fastest executor there (100 rows 8.2 vs 10.9 ms unfused,
     *                           1 536 rows 10.9 vs 14.2, 3 500 rows 17.0 vs 17.5 on the plain stack kernel).  NOT the default of the
     *                           library: the members of a group wait for each other, so ALL its workgroups must be resident — two
     *                           such decodes on one GPU at the same time (two host threads on two streams, two processes sharing
     *                           the device) can starve each other.  The waits are BOUNDED: a group that never meets raises the
     *                           decode's status word, the launches run to their end without waiting, the outputs become all-pad
     *                           captions with NaN log-probs and ortk_decode_status() returns ORTK_EEXCHANGE — no hang, no wrong
     *                           tokens.  Which exchange a group uses does not depend on where the dispatcher put its members: they
     *                           report their XCD (HW_REG_XCC_ID) in every launch, and only a group seen on ONE XCD exchanges through
     *                           that XCD's L2; any other placement takes write-through (sc1) stores — same results, ~1 us more per
     *                           exchange.  Set the flag when this decode has the GPU to itself (the Python model does unless told
     *                           otherwise);
     *   bits 8-15               measurement / tests only: skip self-attention (1) / cross-attention (2) / the FFN (4), no L2
     *                           prefetchers (8); column-split form: deal the members of a group over different XCDs (16), one member
     *                           of group 0 never arrives (32). */
    int32_t exec_flags;
    /* multinomial only: also decode ONE greedy row per image in the same pass (the SCST baseline of
     * utils/training.py:220-237): K = num_random_sample + 1, row 0 of each image is the arg-max decode, rows 1.. are
     * the samples — token for token what two separate calls return, at half the launches. */
    int32_t with_greedy;
    /* multinomial only: index of this call's first output row in the full batch, so that a host that decodes a batch
     * in several chunks (e.g. one per stream) draws the same tokens as one call would (the Gumbel hash is keyed by row) */
    int64_t sample_row_offset;
    /* multinomial only (num_random_sample > 0): TRAIN-mode sampling — every dropout of the model is on while the captions are
     * drawn (the reference samples its SCST rollouts after model.train(), utils/training.py:224-237), keyed by drop_seed exactly
     * as ortk_forward(train = 1, seed = drop_seed) keys the teacher-forced pass over [BOS, sample]: that pass then reproduces
     * the log-probs of the policy that sampled.  Executors: the column-split stack kernel (mixed precision, <= 4 096 rows,
     * ORTK_DEC_SPLIT_SMALL or ORTK_DEC_STACK_SPLIT: its dropout sites draw the same counter hash) or the unfused executor on fp32
     * caches.  On the former `with_greedy` is served too: row 0 of every image stays an EVAL-mode row — no dropout, attending to the
     * eval-mode encoder memory, which this call computes from att_feats — i.e. the reference's separate greedy baseline
     * (utils/training.py:216-222) rides in the launches of the train-mode rollouts; the other rows of image q draw like rows
     * q * num_random_sample + k - 1 of the teacher-forced pass.  (ortk_decode_workspace_bytes returns 0 for train + with_greedy where
     * only the unfused executor applies: run the two decodes separately there.) */
    int32_t train;
    uint64_t drop_seed;
    /* Optional: the encoder memory of these B images, (B*S, d_model) rows in the activation type of cfg->precision (bf16 in mixed
     * precision, fp32 otherwise), e.g. ortk_train_workspace_memory() after ortk_forward_phase(.., 1, ..).  The decode then skips
     * its own encoder pass (att_feats / boxes may be NULL).  With `train` it must be the TRAIN-mode memory under drop_seed
     * (ortk_forward_phase(train = 1, seed = drop_seed, phase 1)); the eval-mode rows of `with_greedy` still take att_feats / boxes. */
    const void* memory;
} ortk_decode_opts;

size_t ortk_decode_workspace_bytes(const ortk_config* cfg, int32_t B, int32_t S, const ortk_decode_opts* o);
/* seq_out (B,K,seq_len) int64, logprob_out (B,K,seq_len) fp32, score_out (B,K) fp32 or NULL; K = beam | samples | 1. */
int ortk_decode(const ortk_config* cfg, const float* params, const float* att_feats, const float* boxes,
                const float* att_masks, int32_t B, int32_t S, const ortk_decode_opts* o, void* ws, size_t ws_bytes,
                int64_t* seq_out, float* logprob_out, float* score_out, ortk_stream stream);
/* Status of the ortk_decode that last ran on workspace `ws` (waits for `stream`: the one host synchronisation of the decode API):
 * 0, or ORTK_EEXCHANGE (see ORTK_DEC_SPLIT_SMALL; only the column-split stack kernel can fail this way). */
int ortk_decode_status(const void* ws, ortk_stream stream);

/* Encoder only (EncoderDecoder.encode, relation_transformer.py:69-70): memory_out (B,S,d).
 * Workspace: ortk_decode_workspace_bytes(cfg, B, S, {beam_size = 1}). */
int ortk_encode(const ortk_config* cfg, const float* params, const float* att_feats, const float* boxes,
                const float* att_masks, int32_t B, int32_t S, void* ws, size_t ws_bytes, float* memory_out,
                ortk_stream stream);

/* Per-step host API: what RelationTransformerModel.get_logprobs_state (relation_transformer.py:374-387) runs per call.
 * ortk_project_memory: cross_kv (mem_rows, L*2*d) = memory (mem_rows, d) x the stacked src_attn K|V weights of all layers
 *   (in general U*cw columns: U = distinct decoder layers under share_dec; cw = d with share_att_dec "kv", otherwise 2*d)
 *   (the reference fills its src_attn caches on the first step, transformer.py:255-273).
 * ortk_decode_step: tokens it (rows) at position t -> log-probs logp_out (rows, ld_out).  self_k / self_v are
 *   (L, rows, tmax, d) fp32 caches holding positions < t; position t is appended.  `rows / kv_groups` consecutive rows share
 *   the cross_kv (kv_groups*S, L*2*d) and att_masks (kv_groups, S) of one group.
 * Workspace for both: ortk_decode_step_workspace_bytes(cfg, rows). */
size_t ortk_decode_step_workspace_bytes(const ortk_config* cfg, int32_t rows);
int ortk_project_memory(const ortk_config* cfg, const float* params, const float* memory, int64_t mem_rows, void* ws,
                        size_t ws_bytes, float* cross_kv, ortk_stream stream);
int ortk_decode_step(const ortk_config* cfg, const float* params, const int64_t* it, int32_t t, int32_t rows,
                     int32_t kv_groups, int32_t S, const float* cross_kv, const float* att_masks, float* self_k,
                     float* self_v, int32_t tmax, void* ws, size_t ws_bytes, float* logp_out, int64_t ld_out,
                     ortk_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (each is also what the executor calls; exported for parity tests).
 * ---------------------------------------------------------------------------------------------- */

/* C = epilogue(op(A) * op(B)).  Replaces torch.nn.functional.linear and its autograd
 * (transformer.py:238,280,324-325,412; relation_transformer.py:168-176,191,331-333).
 *   transA = 0: A is (M,K) row-major, lda;   1: A is stored (K,M) row-major.
 *   transB = 0: B is (N,K) row-major (a torch Linear weight), ldb;   1: B is stored (K,N) row-major.
 *   v = acc + bias[n]; relu; v *= rowscale[m]; dropout(p, seed, index m*N+n); v *= (gate[m,n]>0)*gate_scale;
 *   v += resid[m,n];  then C = v, or C += v (atomically, K split over `splitk` workgroups) if accumulate. */
typedef struct ortk_gemm_args {
    const void* A; const void* B; void* C;
    int64_t lda, ldb, ldc;
    int32_t M, N, K, transA, transB;
    const float* bias; const float* rowscale; const float* resid; int64_t ldr;
    const void* gate; int64_t ldg; float gate_scale;
    int32_t relu; float drop_p; uint32_t drop_seed;
    int32_t accumulate, splitk, precision;
    int32_t a_dtype, b_dtype, c_dtype, gate_dtype;   /* storage type of A / B / C / gate: 0 = fp32, 1 = bf16 (precision 1 only) */
    float* colsum;   /* optional, transA: colsum[m] += sum_k A[k,m] (bias gradient fused into the wgrad GEMM; precision 0 outside the split kernels' shapes: its own launch) */
    /* dropout draw of output element (m, n): index (m * drop_row_stride + drop_row_off) * N + n; 0 / 0 = the plain m * N + n.
     * A decode step at position t reproduces the draws of the teacher-forced (rows x T positions, N) output with (T, t). */
    int32_t drop_row_stride, drop_row_off;
    /* Row-wise LayerNorm fused into the epilogue (C fp32 with ldc == N <= 2048; no transA / accumulate / gate / rowscale).
     *   ln_mode 1: C = v as above, ln_y = LayerNorm(v) (ln_y_dtype), ln_stats[m] = {mean, std}: the projection + residual +
     *              next sublayer's norm of SublayerConnection (transformer.py:345-358) in one launch.
     *   ln_mode 2: the product (no bias / residual / relu) is dy, the gradient of the LayerNorm output of input rows ln_x with
     *              statistics ln_stats: C = dLN/dx (+ ln_dres), ln_da / ln_db +=, and the optional copy ln_y =
     *              dropout(C; drop_p, drop_seed, index m*N+n) (ortk_layernorm_bwd_drop's dz).
     * One launch for N = 512 in mixed precision with bf16 operands; any other case runs the separate kernels. */
    int32_t ln_mode, ln_y_dtype;
    const float* ln_a; const float* ln_b; void* ln_y; float* ln_stats; float ln_eps;
    const float* ln_x; const float* ln_dres; float* ln_da; float* ln_db;
    /* Soft-max partials of the output rows (the generator of a decode step: the beam step then reads 2 floats per 64 logits
     * instead of the logits): tile_stats[(m * ceil(N / 64) + j) * 2 + {0, 1}] = {max, sum exp(v - max)} over the columns
     * [64 j, 64 j + 64) below stat_ncols of row m of C (bias included; {-inf, 0} for an empty block).  Mixed precision, bf16
     * operands, forward layout, plain bias epilogue, N a multiple of 128, K of 64; ORTK_EINVAL otherwise. */
    float* tile_stats; int32_t stat_ncols;
    /* optional (M ints): output row m takes the dropout draws of row drop_rows[m] — the valid-position decoder layout
     * (ortk_batch.row_pos) draws what the padded (caption, position) layout draws, so a step is the same function of its seed in
     * both layouts and an SCST update on the valid positions reproduces the masks of its train-mode rollout. */
    const int32_t* drop_rows;   /* applied before drop_row_stride / drop_row_off */
    /* Gumbel-max sampling candidates of the output rows (the generator of a sampling decode step, with tile_stats): per row m and block j of
     * 64 columns, tile_samp[(m * ceil(N / 64) + j) * 4 + {0, 1, 2}] = {best key, its column (int bits), the logit at that column} where
     * key(m, v) = C[m, v] * samp_inv_temperature + gumbel(samp_seed, samp_t, hash row of m, v) on sampling rows and C[m, v] on greedy rows
     * (samp_greedy_stride K > 0: rows with (samp_row_offset + m) % K == 0; the hash row of a sampling row is its index among the sampling
     * rows — oracle/ort_oracle.py: gumbel_from_hash, caption_model.py:56-111's multinomial draw as a Gumbel arg-max), over the columns below
     * stat_ncols other than the row's previous token samp_seq[m * samp_L + samp_t - 1] (decoding constraint; NULL: none).  A combine step
     * (the decode executor's) takes the arg-max over the blocks and the token's log-prob from tile_stats.  samp_no_store: C is not written. */
    float* tile_samp; const int64_t* samp_seq; uint64_t samp_seed; int64_t samp_row_offset;
    int32_t samp_L, samp_t, samp_greedy_stride, samp_sample, samp_fast, samp_no_store; float samp_inv_temperature;
} ortk_gemm_args;
int ortk_gemm(const ortk_gemm_args* a, ortk_stream stream);
/* The weight (and bias) gradients of up to ORTK_WGRAD_MAX projections over the SAME `rows` batch rows in one launch (mixed precision):
 *   dW_i (Nout_i x Kin_i, fp32, ld lddw) += dY_i^T X_i,   db_i (Nout_i, optional) += column sums of dY_i,
 * dY_i (rows x Nout_i) and X_i (rows x Kin_i) bf16 row-major, 16-byte aligned, leading dimensions and widths multiples of 8.
 * Replaces the per-Linear weight / bias autograd of the reference's step (scripts/train_transformer.py:65-81 over the nn.Linear
 * modules of models/transformer.py:214-358): a layer's gradients share one grid of 256 x 256 tiles x `splitk` row ranges
 * (0 = automatic: one round of workgroups) that add into the arena with 256-byte atomic rows; any row count.  ORTK_EINVAL for
 * anything else (the caller then runs the same products through ortk_gemm, transA = transB = 1, accumulate). */
#define ORTK_WGRAD_MAX 8
typedef struct ortk_wgrad_item {
    const void* dY; int64_t lddy;
    const void* X; int64_t ldx;
    float* dW; int64_t lddw;
    float* db;
    int32_t Nout, Kin;
} ortk_wgrad_item;
typedef struct ortk_wgrad_group_args {
    ortk_wgrad_item item[ORTK_WGRAD_MAX];
    int32_t n, splitk;
    int64_t rows;
    int32_t flags;      /* 0; measurement only: 1 = both waves of a SIMD in lock step (one barrier per stage) instead of the ping-pong schedule,
                           8 = atomics although a workspace is given */
    void* ws; size_t ws_bytes;   /* optional, 256-byte aligned, ortk_wgrad_group_workspace_bytes(a): the row ranges of a tile then meet in memory
                                    (partial tiles + a ticket per tile; the last range to arrive adds them up with plain loads / stores) instead
                                    of adding to the arena with atomics.  The workspace must not be shared by launches on different streams. */
} ortk_wgrad_group_args;
int ortk_wgrad_group(const ortk_wgrad_group_args* a, ortk_stream stream);
size_t ortk_wgrad_group_workspace_bytes(const ortk_wgrad_group_args* a);
/* Measurement hook (off by default, never on inside a timed region): HIP events around every ortk_gemm launch on its
 * launch stream, summed per kernel instance key = precision*4 + transA*2 + transB.  collect() synchronises.
 * on = 1: the executor also keeps every launch on the caller's stream (kernels timed one at a time); on = 2: the schedule of
 * the timed region as it is (weight-gradient GEMMs on the executor's side stream beside the launches being timed). */
int ortk_prof_enable(int32_t on);
int ortk_prof_collect(int32_t key, int64_t* launches, double* total_ms, double* total_flops);
/* algorithmic bytes of the same launches: operands once (A: M x K, B: N x K), the output once, residual / gate rows once */
int ortk_prof_collect_bytes(int32_t key, double* total_bytes);
/* workgroups started by the launches of `key`, summed (key 18 = ortk_wgrad_group launches: they are sized to hold part of the chip) */
int ortk_prof_collect_units(int32_t key, double* total_workgroups);

/* LayerNorm of transformer.py:338-341: a*(x-mean)/(std_unbiased+eps)+b.  stats (rows,2) = {mean, std}. */
int ortk_layernorm_fwd(const float* x, const float* a, const float* b, void* y, int32_t y_dtype, float* stats,
                       int64_t rows, int32_t d, float eps, ortk_stream stream);
/* dx = dLN/dx (+ dres if non-NULL); da, db accumulate (+=). */
int ortk_layernorm_bwd(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                       float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, ortk_stream stream);
/* Same, with a second output dz (dz_dtype 0 fp32 / 1 bf16) = what ortk_dropout_apply(dx, dz, dz_dtype, rows*d, drop_p,
 * drop_seed) would write: the gradient entering the previous sublayer's out-projection (SublayerConnection dropout,
 * transformer.py:345-358), produced in the same pass instead of a separate read of dx. */
int ortk_layernorm_bwd_drop(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                            float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                            float drop_p, uint32_t drop_seed, ortk_stream stream);
/* Same; drop_rows (optional, `rows` ints): dz[row] takes the draws of row drop_rows[row] (index drop_rows[row]*d + col) — the
 * valid-position decoder layout keyed like the padded one (see ortk_gemm_args.drop_rows). */
int ortk_layernorm_bwd_drop_rows(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                                 float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                                 float drop_p, uint32_t drop_seed, const int32_t* drop_rows, ortk_stream stream);
/* the same with the output gradient dy in fp32 or bf16 (dy_dtype; bf16: d = 512 and 16-byte aligned rows only) */
int ortk_layernorm_bwd_dt(const void* dy, int32_t dy_dtype, const float* x, const float* a, const float* stats, const float* dres,
                          float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                          float drop_p, uint32_t drop_seed, const int32_t* drop_rows, ortk_stream stream);

/* Geometry bias of BoxMultiHeadedAttention (relation_transformer.py:196-256,177-183,286):
 * out[l,b,h,i,j] = log(max(relu(WG[l,h].e_ij + bG[l,h]), 1e-6)).  wg[l]/bg[l] are per-layer device pointers
 * laid out (h,64)/(h).  dim_mat = the 8 fp32 wavelengths 1/1000^(k/8) as torch computes them; dim_mat = NULL selects the
 * non-trigonometric mode (embedding = the 4 log-ratios, WG laid out (h,4); ortk_box_embedding then writes (B,S,S,4)). */
int ortk_box_logbias_fwd(const float* boxes, const float* const* wg, const float* const* bg, const float* dim_mat,
                         float* out, int32_t L, int32_t B, int32_t S, int32_t H, ortk_stream stream);
/* dscore (L,B,H,S,S) -> dwg[l] (H,64) += , dbg[l] (H) += . */
int ortk_box_logbias_bwd(const float* boxes, const float* const* wg, const float* const* bg, const float* dim_mat,
                         const float* dscore, float* const* dwg, float* const* dbg,
                         int32_t L, int32_t B, int32_t S, int32_t H, ortk_stream stream);
/* The 64-d embedding itself, (B,S,S,64) — test/diagnostic only. */
int ortk_box_embedding(const float* boxes, const float* dim_mat, float* out, int32_t B, int32_t S, ortk_stream stream);

/* Scaled-dot-product attention over small tiles held in LDS (transformer.py:285-295, relation_transformer.py:258-293).
 * nkv key/value groups of Lk keys; each group serves Lq consecutive query rows.
 *   score = q.k/sqrt(dk); masked_fill(-1e9) where kmask[g,j]==0 or (causal and j > i % causal_period);
 *   score += bias[g,h,i,j];  P = softmax;  O = dropout(P) V.
 * Q/K/V/O are row matrices with H*dk head-concatenated columns and their own leading dimensions.
 * kv_index (optional, (nkv,Lk) int32) gives the physical K/V row of key j of group g (beam ancestry). */
typedef struct ortk_attn_args {
    const float* q; const float* k; const float* v; void* o;
    int64_t ldq, ldk, ldv, ldo;
    const float* kmask; const float* bias; const int32_t* kv_index;
    int64_t kv_group_stride;        /* K/V rows between consecutive groups when kv_index == NULL (0 = Lk) */
    float* p;                       /* (nkv,H,Lq,Lk) probabilities, saved for backward (may be NULL in fwd) */
    int32_t nkv, H, Lq, Lk, dk, causal_period;
    float drop_p; uint32_t drop_seed;
    /* backward only */
    const float* d_o; void* dq; void* d_k; void* dv; float* dscore;  /* dscore (nkv,H,Lq,Lk) or NULL */
    int64_t lddo, lddq, lddk, lddv;
    int32_t o_dtype, dqkv_dtype;    /* storage type of O (fwd) and of dQ/dK/dV (bwd): 0 = fp32, 1 = bf16 */
    int32_t kv_dtype;               /* forward only: 1 = k / v point to bf16 rows (ldk / ldv in elements, multiples of 8): the
                                     * decode-time caches in mixed precision; served for 1-16 query rows, Lk <= 48, dk = 64 */
    /* forward only, Lq = 1 (cached self-attention, transformer.py:265-269): the K / V of the NEW position (fp32 rows
     * k_new[g*ld_new ..], v_new[..]) become key Lk-1 of group g: they are written to that key's cache row and attended to
     * in the same launch (no separate cache-append pass). */
    const float* k_new; const float* v_new; int64_t ld_new;
    /* backward only: 0 = dQ, dK, dV (and dscore) in one call; 1 then 2 = the same work as two calls — part 1 guarantees dQ
     * (and dscore), part 2 completes dK / dV — so that a host can keep part 2 off its critical path (a kernel that cannot
     * split does everything in part 1 and nothing in part 2). */
    int32_t bwd_part;
    /* 0 = fp32 products (v_mfma_f32_16x16x4_f32: the parity mode); 1 = the operands of the four products are rounded to bf16
     * (fp32 accumulation, fp32 soft-max; served for dk = 64 or 32, Lk <= 128, Lq <= 128 — other shapes, and short query blocks over
     * at most 48 fp32 keys, run the fp32 kernels) */
    int32_t precision;
    /* 1 = q, k, v — and, in the backward, d_o — point to bf16 rows (leading dimensions in elements, multiples of 8; 16-byte
     * aligned): the packed projection outputs (and the out-projection's input gradient) of the mixed-precision training
     * step, which only ever feed these products.  Served by the bf16-operand kernels
     * only (precision = 1, dk = 64 or 32, Lk <= 128, Lq <= 128); anything else returns ORTK_EINVAL. */
    int32_t qkv_dtype;
    /* Ragged query groups (bf16-operand kernels only: precision = 1, qkv_dtype = 1; anything else returns ORTK_EINVAL):
     * group g's query rows (q, o, d_o, dq) are rows [q_off[g*q_off_stride], q_off[(g+1)*q_off_stride]) instead of
     * [g*Lq, (g+1)*Lq) — at most Lq of them (Lq stays the row count of the P / bias / dscore blocks).  kv_ragged = 1: the keys
     * of a group are its own query rows (self-attention: k, v, d_k, dv, kmask indexed like q); 0: keys as without q_off. */
    const int32_t* q_off; int32_t q_off_stride; int32_t kv_ragged;
    /* forward only, drop_p > 0: draw the probability dropout of a DECODE step at position drop_tf_t as the teacher-forced pass
     * over drop_tf_T positions draws it (transformer.py:293-294 in train mode during sampling, utils/training.py:224-237):
     * query row i of group g at key j uses index ((g*H + h) * (Lq * drop_tf_T) + i * drop_tf_T + drop_tf_t) * drop_tf_lk + j —
     * self-attention: Lq = 1, drop_tf_lk = drop_tf_T; cross-attention: Lq = samples of the image, drop_tf_lk = regions.
     * drop_tf_T = 0: the natural index ((g*H + h) * Lq + i) * Lk + j.  Served by the generic kernel only. */
    int32_t drop_tf_T, drop_tf_t, drop_tf_lk;
    /* optional, with q_off (ragged query groups), drop_p > 0: query row r (an index into q) draws the probability dropout of
     * row drop_rows[r] of the UNRAGGED layout — group g's query i becomes i' = drop_rows[q_off[g*q_off_stride] + i] - g*Lq in the
     * index ((g*H + h) * Lq + i') * Lk + j.  Cross-attention on the valid positions (an image's rows are the valid positions of
     * its captions) then draws what the padded layout draws; forward and backward alike. */
    const int32_t* drop_rows;
} ortk_attn_args;
int ortk_attention_fwd(const ortk_attn_args* a, ortk_stream stream);
int ortk_attention_bwd(const ortk_attn_args* a, ortk_stream stream);

/* InputEmbedding + PositionalEncoding (transformer.py:383-401): out[r*T+t] = lut[tok]*sqrt(d) + pe[t0+t], dropout.
 * Also emits keymask[r*T+t] = (tok != pad). */
int ortk_embed_fwd(const int64_t* seq, int64_t seq_stride, const float* lut, const float* pe, float* out, float* keymask,
                   int64_t R, int32_t T, int32_t t0, int32_t d, int32_t pad_id, float drop_p, uint32_t seed, ortk_stream stream);
int ortk_embed_bwd(const int64_t* seq, int64_t seq_stride, const float* dout, float* dlut,
                   int64_t R, int32_t T, int32_t d, float drop_p, uint32_t seed, ortk_stream stream);

/* In-place log_softmax over the first V columns of (rows, ld) (OutputEmbedding, transformer.py:412-413);
 * logits are first multiplied by `scale` (1/temperature). */
int ortk_log_softmax(float* x, int64_t rows, int32_t V, int64_t ld, float scale, ortk_stream stream);
/* Fused cross-entropy on logits (rows, ld): *loss_dev = -sum logp[target]*w/norm (LanguageModelCriterion / RewardCriterion,
 * utils/losses.py:15-43); dlogits (rows, ld_dl; fp32 or bf16; may alias logits when fp32 with ld_dl == ld) <- dLoss/dlogits,
 * zero in the pad columns.  The sum is DETERMINISTIC: every row's term goes to row_loss[row] and one workgroup adds them in a
 * fixed order (no atomics) — the same bits on every run.  row_loss: ortk_xent_scratch_floats(rows) floats of caller scratch. */
int64_t ortk_xent_scratch_floats(int64_t rows);
int ortk_xent_fwd_bwd(const float* logits, const int64_t* targets, int64_t target_stride, int32_t T, const float* weight,
                      const float* norm_dev, float* loss_dev, float* row_loss, int64_t rows, int32_t V, int64_t ld,
                      void* dlogits, int32_t dl_dtype, int64_t ld_dl, ortk_stream stream);
/* log_softmax backward: dlogits = dlogp - exp(logp) * sum_v dlogp, written over (rows, ld_out). */
int ortk_log_softmax_bwd(const float* logp, const float* dlogp, int64_t ld_in, void* dlogits, int32_t dl_dtype,
                         int64_t ld_out, int64_t rows, int32_t V, ortk_stream stream);

/* out[n] += sum_m x[m,n] (bias gradients). */
int ortk_colsum(const void* x, int32_t x_dtype, int64_t ld, float* out, int64_t M, int32_t N, ortk_stream stream);
/* y = gate > 0 ? x*scale : 0 — backward of relu (+dropout) given the saved forward output. */
int ortk_gate_apply(const float* x, const float* gate, void* y, int32_t y_dtype, int64_t n, float scale, ortk_stream stream);
/* y = x * keep(seed,i)/(1-p) — backward of a residual-branch dropout. */
int ortk_dropout_apply(const float* x, void* y, int32_t y_dtype, int64_t n, float p, uint32_t seed, ortk_stream stream);
/* Same over (rows, d) with row r keyed as row drop_rows[r] (index drop_rows[r]*d + col); drop_rows = NULL: ortk_dropout_apply. */
int ortk_dropout_apply_rows(const float* x, void* y, int32_t y_dtype, int64_t rows, int32_t d, float p, uint32_t seed,
                            const int32_t* drop_rows, ortk_stream stream);
/* fp32 -> bf16 copy (the working copy of the weight arena in precision 1). */
int ortk_cast_bf16(const float* x, void* y, int64_t n, ortk_stream stream);
int ortk_fill(float* x, int64_t n, float value, ortk_stream stream);
/* y[r, 0..cols) += x[r, 0..cols): x and y are column blocks of (rows, ld) matrices of `dtype` (0 fp32, 1 bf16) */
int ortk_axpy_cols(const void* x, void* y, int32_t dtype, int64_t ld, int64_t rows, int32_t cols, ortk_stream stream);
/* *out_dev = sum(x[0..n)) in a fixed order (no atomics: bit-identical reruns).  n <= 65 536 needs no scratch (NULL); longer
 * inputs take ortk_sum_scratch_floats(n) floats. */
int64_t ortk_sum_scratch_floats(int64_t n);
int ortk_sum(const float* x, int64_t n, float* scratch, float* out_dev, ortk_stream stream);

/* clip_grad_value_ + Adam (utils/optim.py:116-126,187-191; torch.optim.Adam update rule).
 * bc1 = 1-beta1^t, bc2 = 1-beta2^t computed by the host in double precision. */
int ortk_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float clip, float bc1, float bc2, ortk_stream stream);
/* The same update; the gradient is cleared on the way (`optimizer.zero_grad()` of scripts/train_transformer.py:66 without its
 * own pass over the arena: the next step's weight-gradient GEMMs accumulate into zeros). */
int ortk_adam_clip_zero(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float clip, float bc1, float bc2, ortk_stream stream);

/* MaskMixin.get_masked_weight over the whole arena (pruning/masked_layer.py:84-110, pruning/sampler.py):
 *   mode 0: s = round(sigmoid(m))  (supermask, eval)      mode 1: s = bernoulli(sigmoid(m)) (supermask, train)
 *   mode 2: s = m                  (binary masks: magnitude / SNIP / mask_freeze)
 * w_eff = s * w.  Backward (straight-through, sampler.py:10-34): dw = dw_eff*s; dm += dw_eff*w*sigmoid'(m)
 * (mode 2: dm += dw_eff*w, used by SNIP only; dm may be NULL).  extra_coef_dev (device scalar or NULL) is added to
 * dLoss/ds of every element before the sigmoid derivative: the gradient of compute_sparsity_loss (prune.py:228-269). */
int ortk_mask_apply(const float* w, const float* m, float* w_eff, int64_t n, int32_t mode, uint32_t seed, ortk_stream stream);
int ortk_mask_bwd(const float* dw_eff, const float* w, const float* m, float* dw, float* dm, int64_t n, int32_t mode,
                  uint32_t seed, const float* extra_coef_dev, ortk_stream stream);
/* The element-wise tail of a masked (supermask / binary-mask) training step over a RANGE of the arena in ONE pass
 * (scripts/train_n_prune_transformer.py:132-168 after loss.backward(); pruning/sampler.py:10-66; utils/optim.py:116-126,187-191):
 *   s = sample(ml) as the forward drew it (mode, seed, draws: see ortk_mask_apply); dW = g * s;
 *   ds = (g * w + extra_coef[0]) * sigmoid'(ml) [mode != 2] * active;           (straight-through + sparsity-loss term; frozen scopes)
 *   clip + Adam(lr_w, eps_w) on w with dW;  clip + Adam(lr_m, eps_m) on ml with ds (mm / mv NULL: the masks are not trained);  g = 0.
 * Every pointer addresses element index0 of its arena (index0 a multiple of 4); bc1 / bc2 = 1 - beta^t as in ortk_adam_clip.
 * 14-15 array passes instead of the 22 of ortk_mask_bwd + a cleared dm + two ortk_adam_clip launches. */
typedef struct ortk_masked_adam_args {
    float *w, *g, *mw, *vw;               /* weights, their gradient (dLoss/d(s*w) in, zeros out), Adam moments */
    float *ml, *mm, *mv;                  /* mask logits (or binary masks, mode 2) and their Adam moments (NULL, NULL: read only) */
    const float* draws;                   /* optional explicit uniforms (mode 1), element i of the range */
    const float* active;                  /* optional 0 / 1 per element: mask logits outside it keep a zero gradient */
    const float* extra_coef;              /* optional device scalar: d(sparsity loss)/d(sample) */
    int64_t n, index0; int32_t mode; uint32_t seed;
    float lr_w, eps_w, lr_m, eps_m, beta1, beta2, clip, bc1, bc2;
} ortk_masked_adam_args;
int ortk_masked_adam_step(const ortk_masked_adam_args* a, ortk_stream stream);
/* Mode 1 with EXPLICIT uniforms draws[i] in [0,1) in place of the counter hash: s = draws[i] < sigmoid(m[i]), torch.bernoulli's
 * definition for a given uniform (pruning/sampler.py:10-17) — lets a caller (and the parity tests, with the reference's own
 * draws) reproduce a given Bernoulli mask sample in the forward and in the straight-through backward. */
int ortk_mask_apply_draws(const float* w, const float* m, const float* draws, float* w_eff, int64_t n, ortk_stream stream);
int ortk_mask_bwd_draws(const float* dw_eff, const float* w, const float* m, const float* draws, float* dw, float* dm, int64_t n,
                        const float* extra_coef_dev, ortk_stream stream);
/* The weight blocks the executor multiplies by (packed Q|K|V, the all-layer cross-attention K|V block, ...): block i
 * is the (N,K) row-major matrix at arena offset *offset.  Returns the number of blocks when i < 0. */
int ortk_linear_block(const ortk_config* cfg, int32_t i, int64_t* offset, int32_t* N, int32_t* K);
/* count_dev[0] += number of kept entries (round(sigmoid(m)) for mode 0/1, m != 0 for mode 2) in m[0..n). */
int ortk_mask_count(const float* m, int64_t n, int32_t mode, float* count_dev, ortk_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* ORTK_H */
