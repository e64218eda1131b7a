// Internal (non-ABI) launchers shared between the decode kernels and the executor.
#pragma once
#include "ortk_common.h"

namespace ortk {

// process-wide A/B switches (ortk.h: ortk_tuning; defaults = the product path).  Set explicitly through ortk_set_tuning — nothing in
// the library reads the environment.
const ortk_tuning& tuning();
// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for `fn` on the current device: set once per (device, function), thread-safe
int lds_attr(const void* fn, size_t bytes);

// true while bench.py's per-launch GEMM timing is on (the executor then keeps every GEMM on the caller's stream)
bool ortk_prof_active();
// ortk_prof_enable(1): additionally, the executor keeps everything on ONE stream (each kernel timed alone); (2): the timed
// schedule as it is (side stream on: the events of a launch then include what runs beside it)
bool ortk_prof_serial();
// the same measurement hook around a launch that is not an ortk_gemm; ortk_prof_collect(key) then reports it
constexpr int PROF_KEY_DECSTACK = 16;
constexpr int PROF_KEY_CHAIN = 17;        // row_chain_kernel launches (ortk_chain.hip)
constexpr int PROF_KEY_WGRAD_GROUP = 18;  // wgrad_group_kernel launches (ortk_wgrad.hip)
struct ProfMark { hipEvent_t a, b; int key; double flops, bytes; bool live; int slot = -1; double per_count = 0.0; double units = 0.0 /* workgroups of the launch (ortk_prof_collect_units) */; };
bool prof_begin(int key, double flops, double bytes, hipStream_t s, ProfMark& m);
void prof_end(const ProfMark& m, hipStream_t s);
// A device counter a launch's algorithmic bytes depend on (the beam step counts the UNIQUE cache rows of the next decoder pass into it:
// beams share ancestors through the ancestry table): ortk_prof_collect_bytes adds ProfMark.per_count x the counter's final value.
// nullptr (and *index = -1) when profiling is off or the slots are used up.
unsigned long long* prof_slot(int* index);

// copy the new token's K and V (columns d..3d of the packed QKV row) into the self-attention cache
int kv_append(const float* qkv, void* cache_k, void* cache_v, int32_t kv_dtype, int64_t rows, int32_t d, int32_t row_mult, int32_t tmax,
              int32_t t, hipStream_t s);

// bf16-MFMA attention (ortk_attn16.hip): whether the call's shapes / layouts are served, and the launchers
bool attn16_shape_ok(int Lq, int Lk, int dk);
bool attn16_ok(const ortk_attn_args* a, bool bwd);
int attn16_fwd(const ortk_attn_args* a, hipStream_t s);
int attn16_bwd(const ortk_attn_args* a, hipStream_t s);

// bf16 TRANSPOSED copies of weight blocks: block i is an (N, K) row-major fp32 matrix at x + off; yt + off receives it as
// (K, N) row-major bf16 (the data-gradient GEMM dX = dY W then runs in the forward operand layout, on the LDS-DMA kernels)
struct WBlock { int32_t off, N, K, tile0; };      // tile0: index of the block's first 64 x 64 tile in the launch
constexpr int MAX_WBLOCKS = 200;
struct WBlockTable { int32_t n, tiles; WBlock b[MAX_WBLOCKS]; };
int cast_bf16_transposed(const float* x, void* yt, const WBlockTable& t, hipStream_t s);

struct BeamState {
    int32_t B, b, L, V, eos;
    int64_t ldv;
    // ping-pong histories, (B*b, L)
    int32_t* seq[2];
    float* tok_lp[2];
    float* cum;               // (B*b)
    int64_t* it;              // (B*b) tokens fed to the next decoder pass
    int32_t* kvidx[2];        // (B*b, t+1) physical cache rows of every ancestor key
    // finished hypotheses, capacity b*L per image
    int32_t* done_seq;        // (B, b*L, L)
    float* done_lp;           // (B, b*L, L)
    double* done_p;           // (B, b*L)   (penalised score; double like the reference's python floats)
    int32_t* done_len;        // (B, b*L)
    int32_t* done_cnt;        // (B)
    int32_t decoding_constraint, length_penalty;
    double length_alpha;
    int32_t tmax;             // cache time capacity
    // optional soft-max partials of the logit rows (ortk_gemm_args.tile_stats): {max, sum exp} per block of 64 columns
    const float* gstats; int32_t nblk;
    unsigned long long* uniq = nullptr;   // measurement only (prof_slot): += the distinct cache rows the images' new beams reference at positions 0..t
};
// fused = true: `logp` holds raw logits and the log-soft-max of (logits * scale) is taken inside the step
int beam_step(const BeamState& st, const float* logp, int32_t t, hipStream_t s, bool fused = false, float scale = 1.f, bool fast_exp = false);
int beam_finalize(const BeamState& st, int64_t* seq_out, float* lp_out, float* score_out, hipStream_t s);

struct SampleState {
    int32_t rows, L, V, eos;
    int64_t ldv;
    int64_t* it;              // (rows) next tokens
    int64_t* seq;             // (rows, L)   output
    float* lp;                // (rows, L)   output
    int32_t* unfinished;      // (rows)
    int32_t* last_step;       // (1) max over rows of the step at which the row finished
    int32_t decoding_constraint, sample;
    float temperature;
    uint64_t seed;
    int32_t greedy_stride;    // K > 0: rows with row % K == 0 decode greedily (last_step[1]); the others sample (last_step[0])
    int64_t row_offset;       // index of row 0 in the full batch (chunked decoding): the Gumbel hash is keyed by the global row
};
int sample_init(const SampleState& st, int32_t bos, hipStream_t s);
// fused = true: `logp` holds raw logits (V <= 10 240) and the log-soft-max is taken inside the step
int sample_step(const SampleState& st, const float* logp, int32_t t, hipStream_t s, bool fused = false, bool fast_exp = false);
// the same step from the generator GEMM's sampling epilogue (ortk_gemm_args.tile_stats + tile_samp): the logit rows are never materialised
int sample_combine(const SampleState& st, const float* gstats, const float* gsamp, int32_t nblk, int32_t t, hipStream_t s, bool fast_exp);
int sample_finalize(const SampleState& st, hipStream_t s);

// One decode position of the whole decoder stack in one launch (ortk_decstack.hip; mixed precision, d_model 512, 8 heads,
// d_ff a multiple of 512, bf16 caches).
constexpr int STACK_MAXL = 8;
struct StackLayer {
    const float *n0a, *n0b, *bqkv, *bo, *n1a, *n1b, *cqb, *cob, *n2a, *n2b, *b1, *b2;
    __bf16 *ck, *cv;               // self-attention caches, (cache rows, 512)
    const __bf16 *xk, *xv;         // this layer's K / V columns of the projected memory, row pitch ldx
    const __bf16 *xkg, *xvg;       // train-mode decode with greedy rows: the same columns of the EVAL-mode memory's projection (or NULL)
};
struct StackArgs {
    StackLayer layer[STACK_MAXL];
    const uint4* wpk;              // stack_pack() image of the decoder weights (dense stream)
    const uint2* sstream;          // sparse stream (sstack_pack): per wave, steps of 64 lanes x 2 scatter entries
    int32_t gather;                // the sparse stream holds per-column gather lists (gstack_pack) instead: decoder_stack_kernel<true, 20, true>
    const int32_t* snst;           //   [8][L * U] steps of every (wave, unit) (multiples of 4)
    const int64_t* sstart;         //   [8] first step of every wave's stream
    const float* x_io;             // (rows, 512) embedded tokens of this position — or, with `tok`, not read:
    const int64_t* tok; const float* lut; const float* pe_t; float emb_scale;      // x = lut[tok[row]] * emb_scale + pe_t (eval mode: the embedding launch folded into the kernel's first load)
    __bf16* y_out;                 // (rows, 512) final LayerNorm output: the generator's operand
    const float *fa, *fb;          // final LayerNorm
    const float* att_masks;        // (images, S)
    const int32_t* kvidx;          // (rows, t + 1) beam ancestry (cache rows) or NULL: row g owns cache rows g T .. g T + t
    int64_t ldx;
    int32_t rows, per_img, S, T, t, L, NC;
    float eps;
    int32_t nblocks;               // compute workgroups (set by stack_step); workgroups beyond are L2 prefetchers
    int32_t* progress;             // [16] zeroed at the start of a decode: units begun by the pace-maker of each XCD
    int32_t uniq_slot;             // measurement only: prof_slot index + 1 of the counter of unique cache rows this pass references (0: none)
    int32_t debug;                 // measurement / test only: 1 skip self-attention, 2 skip cross-attention, 4 skip the FFN, 8 no L2
                                   // prefetchers; column-split form: 16 deal the members of a group over DIFFERENT XCDs (the placement check
                                   // must then pick the write-through exchange), 32 one member of group 0 never arrives (the bounded wait
                                   // must report ORTK_EEXCHANGE instead of hanging)
    int32_t rb;                    // rows per workgroup: 32 (default) or 20 (256 workgroups for 1 024 images x 5 beams)
    // column-split form (stack_tp_step): G workgroups of one XCD share 64 rows, each owns 512 / G output columns of every unit
    int32_t tp;                    // G: 0 (off) | 2 | 4 | 8
    const uint4* tp_wpk;           // stack_tp_pack() image of the decoder weights
    char* tp_xbuf;                 // exchange tiles: (groups, 2, 64 x 512 x 4 bytes)
    int32_t* tp_flag;              // (groups, TP_FLAG_STRIDE) per group: [0] the exchange counter, [16 + launch] the XCC ids its members
                                   // reported in that launch (bit mask); zeroed at the start of a decode
    int32_t* tp_status;            // [1] decode-wide error word (0 = fine; TP_ERR_*), zeroed at the start of a decode
    int32_t tp_groups;             // groups of the FULL row count of the decode (a multiple of 8); this launch may use fewer
    int32_t tp_launch;             // index of this launch within the decode (the counters keep running)
    int64_t tp_xtile;              // bytes of one exchange tile (stack_tp_xtile_bytes)
    // train-mode rows (column-split form, G >= 4; ortk_decode_opts.train): every dropout of the decoder draws what the teacher-forced
    // pass of the same seed draws at (row, position t) — utils/training.py:224-237 samples after model.train()
    float drop_p;                  // 0: eval mode
    int32_t greedy_stride;         // K > 0: rows with row % K == 0 are EVAL-mode rows (the greedy baseline: no dropout, memory xkg / xvg);
                                   // the other rows of image q are rows q (K - 1) + k - 1 of the teacher-forced pass
    uint32_t drop_seed[STACK_MAXL][6];   // site seeds of layer l: self-attention probabilities, wo output, cross-attention probabilities, co
                                         // output, FFN hidden units, w2 output (ortk_dropout_site_seed(seed, 3, l, k))
};
constexpr int TP_FLAG_STRIDE = 128;        // ints per group in StackArgs.tp_flag
constexpr int TP_ERR_TIMEOUT = 1;          // a member of an exchange group did not arrive within the spin bound
struct StackPack { int64_t off[STACK_MAXL][6]; int32_t L, NC; };   // element offsets of wqkv, wo, cqw, cow, w1, w2 per layer
size_t stack_packed_bytes(int L, int NC);
int stack_pack(const void* w16, void* wpk, const StackPack& t, hipStream_t s);
int stack_step(const StackArgs& a, hipStream_t s);
// column-split form: G (0 = not served), weight image bytes, exchange buffer bytes, flag ints for a decode of `rows` rows
int stack_tp_degree(int64_t rows);
size_t stack_tp_packed_bytes(int L, int NC, int G);
int stack_tp_pack(const void* w16, void* wpk, const StackPack& t, int G, hipStream_t s);
inline int stack_tp_groups(int64_t rows) { return (int)(((rows + 63) / 64 + 7) / 8 * 8); }
// one exchange tile: [64 x 512] fp32, or the NC hidden chunks of [64 x 512] bf16 that travel together
inline size_t stack_tp_xtile_bytes(int NC) { const size_t h = (size_t)NC * 64 * 512 * 2, f = (size_t)64 * 512 * 4; return h > f ? h : f; }
inline size_t stack_tp_xbuf_bytes(int64_t rows, int NC) { return (size_t)stack_tp_groups(rows) * 2 * stack_tp_xtile_bytes(NC); }
// Sparse stream (mostly-zero decoder weights): the non-zeros of every 2 048-weight MFMA fragment group as (position, value) scatter
// entries, built on the device from the bf16 weights of the call (three launches, no host sync).
struct SStackBufs { uint2* stream; int32_t *cnt, *nst; int64_t *start, *stats; size_t stream_bytes; };     // stats: {steps, non-zeros}   // stats: {steps, pieces}
size_t sstack_bytes(int L, int NC, SStackBufs* carve, void* base);          // worst-case (fully dense weights) capacity
int sstack_pack(const void* w16, const SStackBufs& b, const StackPack& t, hipStream_t s);
int gstack_pack(const void* w16, const SStackBufs& b, const StackPack& t, hipStream_t s);       // gather lists in the same buffers

// embedding / fused criterion over an explicit list of (caption, position) rows (row_pos[i] = r*T + t; NULL = all R*T rows in order):
// the valid-position decoder layout of ortk_batch.cap_off / row_pos
int embed_fwd_rows(const int64_t* seq, int64_t seq_stride, const float* lut, const float* pe, float* out, float* keymask, int64_t nrows,
                   const int32_t* row_pos, int32_t T, int32_t t0, int32_t d, int32_t pad_id, float drop_p, uint32_t seed, hipStream_t s,
                   int32_t drop_rs = 1, int32_t drop_r0 = 0, int32_t eval_stride = 0);
int embed_bwd_rows(const int64_t* seq, int64_t seq_stride, const float* dout, float* dlut, int64_t nrows, const int32_t* row_pos, int32_t T,
                   int32_t d, float drop_p, uint32_t seed, hipStream_t s);
int xent_rows(const float* logits, const int64_t* targets, int64_t target_stride, int32_t T, const float* weight, const float* norm_dev,
              float* loss_dev, float* row_loss, int64_t rows, const int32_t* row_pos, int32_t V, int64_t ld, void* dlogits, int32_t dl_dtype,
              int64_t ld_dl, hipStream_t s);
int64_t xent_scratch_floats(int64_t rows);
int64_t sum_partials(int64_t n);
int sum_fixed(const float* x, int64_t n, float* part, float* out_dev, hipStream_t s);

// end of a decode on the column-split stack kernel: if *status != 0 (an exchange group never completed) the outputs become
// all-pad captions with NaN log-probs and scores — a failed decode cannot pass for a result
int decode_poison(const int32_t* status, int64_t* seq, float* lp, float* score, int64_t nseq, int64_t nscore, hipStream_t s);
// Rows-stationary chains of row-wise operators (ortk_chain.hip): pack the 512 x 512 weight units of a chain (bf16, device table of
// descriptors, stream order) into the streaming layout, and run a chain on a packed stream
size_t chain_packed_bytes(int n_units);
int chain_rows_per_block(int64_t M, int slots);
int chain_pack(const void* w16, const ortk_chain_unit* units_dev, int n_units, void* packed, hipStream_t s, bool wide = false, int n_r = 0, int n1 = 0, int NC = 0);
// all chains of a model in ONE launch: chain i = units [first[i], first[i + 1]) of the table, packed stream at uint4 index base[i]
constexpr int CH_MAX_UNITS = 192, CH_MAX_CHAINS = 24;
struct ChainPackTable {
    int32_t n_chains; int32_t first[CH_MAX_CHAINS + 1]; int64_t base[CH_MAX_CHAINS];
    struct { int32_t offset, ld; } u[CH_MAX_UNITS];      // element offset of the unit's first output row in the bf16 arena, leading dimension
    uint8_t form[CH_MAX_UNITS];                          // 0: 8-wave stream | 1: 4-wave (wide) stream | 2: wide, FFN up-projection (two half units)
};
// the 76-row form of the forward chain kernel serves M rows in fewer rounds of workgroups than the 48-row form (and is switched on)
bool chain_wide(int64_t M);
// form of unit `u` of a forward chain with `n_r` (0 | 1) leading out-projection units, n1 projections behind the first LayerNorm and NC FFN chunks
__host__ __device__ inline uint8_t chain_unit_form(bool wide, int u, int n_r, int n1, int NC) {
    if (!wide) return 0;
    const int k = u - n_r - n1;
    return (k >= 0 && k < 2 * NC && !(k & 1)) ? 2 : 1;
}
int chain_pack_all(const void* w16, void* packed, const ChainPackTable& t, hipStream_t s);
int chain_run(const ortk_chain_args* p, const void* packed, hipStream_t s);
int bchain_run(const ortk_bchain_args* p, const void* packed, hipStream_t s);
int fill_i64(int64_t* p, int64_t n, int64_t v, hipStream_t s);
int fill_i32(int32_t* p, int64_t n, int32_t v, hipStream_t s);
// kvidx[g*1 + 0] = g*row_mult*tmax  (index table for the first decoder pass)
int kvidx_init(int32_t* kvidx, int64_t rows, int32_t row_mult, int32_t tmax, hipStream_t s);

}  // namespace ortk
