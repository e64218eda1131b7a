// ortk_embed_loss.hip — token embedding, generator log-softmax, fused cross-entropy, small reductions.
//
// Replaces InputEmbedding + PositionalEncoding (models/transformer.py:383-401), OutputEmbedding's log_softmax
// (transformer.py:412-413) and LanguageModelCriterion / RewardCriterion (utils/losses.py:15-43).
#include "ortk_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(128) void embed_fwd_kernel(const int64_t* __restrict__ seq, int64_t seq_stride,
                                                        const float* __restrict__ lut, const float* __restrict__ pe,
                                                        float* __restrict__ out, float* __restrict__ keymask, int T, int t0,
                                                        int d, int pad_id, float scale, float drop_p, uint32_t seed,
                                                        const int32_t* __restrict__ row_pos, int drop_rs, int drop_r0, int eval_stride) {
    const int64_t row = blockIdx.x;           // output row; r*T + t, or row_pos[row] in the valid-position layout
    const int64_t prow = row_pos ? (int64_t)row_pos[row] : row;
    const int64_t r = prow / T;
    const int t = (int)(prow - r * T);
    const int64_t tok = seq[r * seq_stride + t];
    if (keymask && threadIdx.x == 0) keymask[row] = tok != pad_id ? 1.f : 0.f;
    const float* e = lut + tok * d;
    const float* p = pe + (int64_t)(t0 + t) * d;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // eval_stride K > 0 (train-mode decode with greedy rows): rows with row % K == 0 are eval-mode rows — no dropout — and row
    // q K + k (k >= 1) draws like row q (K - 1) + k - 1
    int64_t drow = prow;      // (the valid-position layout draws what the padded (caption, position) layout draws)
    bool drop = drop_p > 0.f;
    if (eval_stride > 0) { const int64_t q = row / eval_stride, k = row - q * eval_stride; drop = drop && k != 0; drow = q * (eval_stride - 1) + k - 1; }
    for (int c = threadIdx.x; c < d; c += 128) {
        float v = e[c] * scale + p[c];
        // (draw of output row `row`: its own index, or — a decode step in train mode — that of row*drop_rs + drop_r0 of the
        // teacher-forced pass)
        if (drop) v = ortk_keep(seed, (uint64_t)(drow * drop_rs + drop_r0) * d + c, drop_p) ? v * inv_keep : 0.f;
        out[row * d + c] = v;
    }
}

__global__ __launch_bounds__(128) void embed_bwd_kernel(const int64_t* __restrict__ seq, int64_t seq_stride,
                                                        const float* __restrict__ dout, float* __restrict__ dlut, int T,
                                                        int d, float scale, float drop_p, uint32_t seed,
                                                        const int32_t* __restrict__ row_pos) {
    const int64_t row = blockIdx.x;
    const int64_t prow = row_pos ? (int64_t)row_pos[row] : row;
    const int64_t r = prow / T;
    const int t = (int)(prow - r * T);
    const int64_t tok = seq[r * seq_stride + t];
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int c = threadIdx.x; c < d; c += 128) {
        float g = dout[row * d + c];
        if (drop_p > 0.f) g = ortk_keep(seed, (uint64_t)prow * d + c, drop_p) ? g * inv_keep : 0.f;
        atomicAdd(&dlut[tok * d + c], g * scale);
    }
}

// ---------------------------------------------------------------------------------------------- block reductions
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;
}

// ---------------------------------------------------------------------------------------------- log-softmax / CE
__global__ __launch_bounds__(256) void log_softmax_kernel(float* __restrict__ x, int V, int64_t ld, float scale) {
    __shared__ float sh[4];
    float* row = x + (int64_t)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, row[c] * scale);
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(row[c] * scale - mx);
    s = block_sum(s, sh);
    const float lse = logf(s);
    for (int c = threadIdx.x; c < V; c += 256) row[c] = (row[c] * scale - mx) - lse;
}

__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                   int64_t target_stride, int T, const float* __restrict__ weight,
                                                   const float* __restrict__ norm_dev, float* __restrict__ row_loss, int V,
                                                   int64_t ld, void* dlogits, int dl_dt, int64_t ld_dl, const int32_t* __restrict__ row_pos) {
    __shared__ float sh[4];
    const int64_t r = blockIdx.x;
    const float* row = logits + r * ld;
    const int64_t pr = row_pos ? (int64_t)row_pos[r] : r;
    const int64_t tgt = targets[(pr / T) * target_stride + (pr % T)];
    const float w = weight[pr] / norm_dev[0];
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, row[c]);
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(row[c] - mx);
    s = block_sum(s, sh);
    const float lse = logf(s);
    if (threadIdx.x == 0) row_loss[r] = w != 0.f ? -((row[tgt] - mx) - lse) * w : 0.f;      // summed in a fixed order by loss_reduce_kernel
    __syncthreads();
    // dlogits may alias logits (fp32, same ld): every element is read before it is overwritten by the same thread
    for (int c = threadIdx.x; c < (int)ld_dl; c += 256) {
        float g = 0.f;
        if (c < V && w != 0.f) g = (expf((row[c] - mx) - lse) - (c == tgt ? 1.f : 0.f)) * w;
        st_elem(dlogits, r * ld_dl + c, dl_dt, g);
    }
}

// Register-resident form (V <= 256 * NPT): each thread keeps its strided slice of the row, so the 870 MB logits tensor
// is read ONCE instead of three times; max, sum and gradient run in the element order of xent_kernel (bit-identical).
// FAST (mixed precision: the gradient leaves as bf16): ONE v_exp_f32 per element — e = exp(z - max) stays in the register that held
// z, the gradient is e / sum — instead of two evaluations of libm's expf (~20 vector instructions each: the kernel was bound by them,
// not by its one read of the logits).  The fp32 parity mode keeps the exact form (bit-identical to xent_kernel).
template <int NPT, bool FAST>
__global__ __launch_bounds__(256) void xent_reg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                       int64_t target_stride, int T, const float* __restrict__ weight,
                                                       const float* __restrict__ norm_dev, float* __restrict__ row_loss, int V,
                                                       int64_t ld, void* dlogits, int dl_dt, int64_t ld_dl, const int32_t* __restrict__ row_pos) {
    __shared__ float sh[4];
    const int64_t r = blockIdx.x;
    const float* row = logits + r * ld;
    const int tid = threadIdx.x;
    float z[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) { const int c = tid + 256 * u; z[u] = c < V ? row[c] : 0.f; }
    const int64_t pr = row_pos ? (int64_t)row_pos[r] : r;      // (caption, position) of this row: targets / weights stay (R, T)
    const int64_t tgt = targets[(pr / T) * target_stride + (pr % T)];
    const float w = weight[pr] / norm_dev[0];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NPT; ++u) if (tid + 256 * u < V) mx = fmaxf(mx, z[u]);
    mx = block_max(mx, sh);
    float s = 0.f;
    if (FAST) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) { z[u] = tid + 256 * u < V ? __expf(z[u] - mx) : 0.f; s += z[u]; }
    } else {
#pragma unroll
        for (int u = 0; u < NPT; ++u) if (tid + 256 * u < V) s += expf(z[u] - mx);
    }
    s = block_sum(s, sh);
    const float lse = logf(s);
    if (tid == 0) row_loss[r] = w != 0.f ? -((row[tgt] - mx) - lse) * w : 0.f;
    __syncthreads();     // dlogits may alias logits (fp32 mode): row[tgt] is read before any element is overwritten
    const float ws_ = FAST ? w / s : 0.f;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int c = tid + 256 * u;
        if (c < (int)ld_dl) {
            float g = 0.f;
            if (FAST) { if (c < V && w != 0.f) g = z[u] * ws_ - (c == tgt ? w : 0.f); }
            else if (c < V && w != 0.f) g = (expf((z[u] - mx) - lse) - (c == tgt ? 1.f : 0.f)) * w;
            st_elem(dlogits, r * ld_dl + c, dl_dt, g);
        }
    }
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ dlogp,
                                                              int64_t ld_in, void* __restrict__ dlogits, int dl_dt, int64_t ld_out, int V) {
    __shared__ float sh[4];
    const int64_t r = blockIdx.x;
    const float* lp = logp + r * ld_in;
    const float* dl = dlogp + r * ld_in;
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += dl[c];
    s = block_sum(s, sh);
    for (int c = threadIdx.x; c < (int)ld_out; c += 256) st_elem(dlogits, r * ld_out + c, dl_dt, c < V ? dl[c] - expf(lp[c]) * s : 0.f);
}

// ---------------------------------------------------------------------------------------------- misc
// column sums: each workgroup covers 64 columns x ROWS rows; lanes walk columns (coalesced), waves walk rows.
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int x_dt, int64_t ld, float* __restrict__ out,
                                                     int64_t M, int N) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    float acc = 0.f;
    if (c < N)
        for (int64_t r = r0 + wave; r < min(M, r0 + CS_ROWS); r += 4) acc += ld_elem(x, r * ld + c, x_dt);
    sh[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < N) atomicAdd(&out[c], sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, void* __restrict__ y, int y_dt, int64_t n,
                                                            float p, uint32_t seed) {
    const float inv_keep = 1.f / (1.f - p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        st_elem(y, i, y_dt, (p > 0.f && !ortk_keep(seed, (uint64_t)i, p)) ? 0.f : x[i] * inv_keep);
}

__global__ __launch_bounds__(256) void dropout_apply_rows_kernel(const float* __restrict__ x, void* __restrict__ y, int y_dt, int64_t rows, int d,
                                                                 float p, uint32_t seed, const int32_t* __restrict__ drop_rows) {
    const float inv_keep = 1.f / (1.f - p);
    const int64_t n = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d; const int c = (int)(i - r * d);
        st_elem(y, i, y_dt, (p > 0.f && !ortk_keep(seed, (uint64_t)drop_rows[r] * (uint64_t)d + (uint64_t)c, p)) ? 0.f : x[i] * inv_keep);
    }
}

__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                         void* __restrict__ y, int y_dt, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        st_elem(y, i, y_dt, gate[i] > 0.f ? x[i] * scale : 0.f);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<bf16x4*>(y)[i] = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (__bf16)x[i];
}

// one 64 x 64 tile of one weight block per workgroup: fp32 (N, K) -> bf16 (K, N) through a padded LDS tile
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float* __restrict__ x, __bf16* __restrict__ yt, ortk::WBlockTable tab) {
    __shared__ float tile[64][65];
    int bi = 0;
    {   // the block that owns this tile (tile0 is ascending): binary search
        int lo = 0, hi = tab.n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab.b[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        bi = lo;
    }
    const ortk::WBlock wb = tab.b[bi];
    const int tk = (wb.K + 63) >> 6;
    const int t = (int)blockIdx.x - wb.tile0, n0 = (t / tk) * 64, k0 = (t % tk) * 64;
    const float* src = x + wb.off;
    __bf16* dst = yt + wb.off;
    const int tid = threadIdx.x;
    const bool vec = (wb.K & 3) == 0 && (wb.off & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
        const int n = n0 + r, k = k0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < wb.N) {
            if (vec && k + 3 < wb.K) v = *reinterpret_cast<const float4*>(src + (int64_t)n * wb.K + k);
            else { float* q = &v.x; for (int e = 0; e < 4; ++e) if (k + e < wb.K) q[e] = src[(int64_t)n * wb.K + k + e]; }
        }
        tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
    }
    __syncthreads();
    const bool vec_o = (wb.N & 7) == 0 && (wb.off & 7) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kr = (tid >> 3) + 32 * i, c = (tid & 7) * 8;
        const int k = k0 + kr, n = n0 + c;
        if (k >= wb.K) continue;
        if (vec_o && n + 7 < wb.N) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)tile[c + e][kr];
            *reinterpret_cast<bf16x8*>(dst + (int64_t)k * wb.N + n) = o;
        } else {
            for (int e = 0; e < 8; ++e) if (n + e < wb.N) dst[(int64_t)k * wb.N + n + e] = (__bf16)tile[c + e][kr];
        }
    }
}

__global__ __launch_bounds__(256) void add_cols_kernel(const void* __restrict__ x, void* __restrict__ y, int dt, int64_t ld, int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        st_elem(y, r * ld + c, dt, ld_elem(y, r * ld + c, dt) + ld_elem(x, r * ld + c, dt));
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

// Fixed-order sums (no atomics: the same bits on every run, whatever the block scheduling).  Stage 1: block b sums its contiguous
// SUM_CHUNK-element slice — thread t the elements t, t + 1024, ... in that order, then a tree over the 1 024 partials — into part[b];
// stage 2 (one block) does the same over the partials.  n <= SUM_CHUNK: one launch.  The scalar every caller reads — the loss of
// LanguageModelCriterion / RewardCriterion (utils/losses.py:15-43: one torch.sum over the (rows, T) terms) — comes out of here.
constexpr int SUM_THREADS = 1024;
constexpr int64_t SUM_CHUNK = 64 * 1024;
__device__ __forceinline__ float fixed_order_block_sum(const float* __restrict__ x, int64_t n, float* sh) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += SUM_THREADS) s += x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int k = SUM_THREADS / 2; k >= 1; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    return sh[0];
}
__global__ __launch_bounds__(SUM_THREADS) void sum_fixed_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float sh[SUM_THREADS];
    const int64_t i0 = (int64_t)blockIdx.x * SUM_CHUNK;
    const float s = fixed_order_block_sum(x + i0, n - i0 < SUM_CHUNK ? n - i0 : SUM_CHUNK, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// Tables of the valid-position decoder layout (ortk_batch.cap_off / row_pos) from the per-caption position counts: one workgroup
// scans the counts (R <= 1024 * 64: every thread a contiguous run of captions, a tree over the thread totals), then every caption
// writes the padded (caption, position) index of each of its compact rows.
__global__ __launch_bounds__(1024) void valid_rows_scan_kernel(const int64_t* __restrict__ n, int R, int32_t* __restrict__ cap_off) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (R + 1023) / 1024, r0 = tid * per, r1 = min(R, r0 + per);
    int s = 0;
    for (int r = r0; r < r1; ++r) s += (int)n[r];
    part[tid] = s;
    __syncthreads();
    for (int k = 1; k < 1024; k <<= 1) {          // inclusive scan of the thread totals
        const int v = tid >= k ? part[tid - k] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int acc = tid ? part[tid - 1] : 0;
    if (tid == 0) cap_off[0] = 0;
    for (int r = r0; r < r1; ++r) { acc += (int)n[r]; cap_off[r + 1] = acc; }
}
__global__ __launch_bounds__(256) void valid_rows_fill_kernel(const int32_t* __restrict__ cap_off, int R, int T, int32_t* __restrict__ row_pos) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const int o = cap_off[r], cnt = cap_off[r + 1] - o;
    for (int t = lane; t < cnt; t += 64) row_pos[o + t] = r * T + t;
}

inline unsigned ew_grid(int64_t n) { return (unsigned)std::min<int64_t>(ortk_cdiv(n, 256), 2048); }

}  // namespace

// (nrows output rows; row_pos == NULL: nrows = R * T rows in (caption, position) order)
namespace ortk {
int embed_fwd_rows(const int64_t* seq, int64_t seq_stride, const float* lut, const float* pe, float* out, float* keymask, int64_t nrows,
                   const int32_t* row_pos, int32_t T, int32_t t0, int32_t d, int32_t pad_id, float drop_p, uint32_t seed, hipStream_t s,
                   int32_t drop_rs, int32_t drop_r0, int32_t eval_stride) {
    if (!seq || !lut || !pe || !out || nrows < 0 || T < 1 || d < 1) return ORTK_EINVAL;
    if (nrows == 0) return 0;
    const float scale = (float)sqrt((double)d);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)nrows), dim3(128), 0, s, seq, seq_stride, lut, pe, out, keymask, T, t0, d, pad_id,
                       scale, drop_p, seed, row_pos, drop_rs > 0 ? drop_rs : 1, drop_r0, eval_stride);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int embed_bwd_rows(const int64_t* seq, int64_t seq_stride, const float* dout, float* dlut, int64_t nrows, const int32_t* row_pos, int32_t T,
                   int32_t d, float drop_p, uint32_t seed, hipStream_t s) {
    if (!seq || !dout || !dlut || nrows < 0 || T < 1 || d < 1) return ORTK_EINVAL;
    if (nrows == 0) return 0;
    const float scale = (float)sqrt((double)d);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)nrows), dim3(128), 0, s, seq, seq_stride, dout, dlut, T, d, scale, drop_p, seed, row_pos);
    ORTK_CHECK_LAUNCH();
    return 0;
}
}  // namespace ortk

extern "C" int ortk_embed_fwd(const int64_t* seq, int64_t seq_stride, const float* lut, const float* pe, float* out,
                              float* keymask, int64_t R, int32_t T, int32_t t0, int32_t d, int32_t pad_id, float drop_p,
                              uint32_t seed, ortk_stream stream) {
    if (R < 0) return ORTK_EINVAL;
    return ortk::embed_fwd_rows(seq, seq_stride, lut, pe, out, keymask, R * T, nullptr, T, t0, d, pad_id, drop_p, seed, ortk_s(stream));
}

extern "C" int ortk_embed_bwd(const int64_t* seq, int64_t seq_stride, const float* dout, float* dlut, int64_t R, int32_t T,
                              int32_t d, float drop_p, uint32_t seed, ortk_stream stream) {
    if (R < 0) return ORTK_EINVAL;
    return ortk::embed_bwd_rows(seq, seq_stride, dout, dlut, R * T, nullptr, T, d, drop_p, seed, ortk_s(stream));
}

extern "C" int ortk_log_softmax(float* x, int64_t rows, int32_t V, int64_t ld, float scale, ortk_stream stream) {
    if (!x || rows < 0 || V < 1 || ld < V) return ORTK_EINVAL;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, ortk_s(stream), x, V, ld, scale);
    ORTK_CHECK_LAUNCH();
    return 0;
}

namespace ortk {
// *out_dev = sum(x[0..n)) in a fixed order; `part` (sum_partials(n) floats of scratch) is needed when n > SUM_CHUNK
int64_t sum_partials(int64_t n) { return n > SUM_CHUNK ? ortk_cdiv(n, SUM_CHUNK) : 0; }
int sum_fixed(const float* x, int64_t n, float* part, float* out_dev, hipStream_t s) {
    if (!x || !out_dev || n < 0) return ORTK_EINVAL;
    if (n == 0) return ortk_fill(out_dev, 1, 0.f, (ortk_stream)s);
    const int64_t nb = ortk_cdiv(n, SUM_CHUNK);
    if (nb > 1 && !part) return ORTK_EINVAL;
    if (nb > SUM_CHUNK) return ORTK_EINVAL;      // (4e9 elements: no caller of this path)
    hipLaunchKernelGGL(sum_fixed_kernel, dim3((unsigned)nb), dim3(SUM_THREADS), 0, s, x, n, nb > 1 ? part : out_dev);
    ORTK_CHECK_LAUNCH();
    if (nb > 1) {
        hipLaunchKernelGGL(sum_fixed_kernel, dim3(1), dim3(SUM_THREADS), 0, s, part, nb, out_dev);
        ORTK_CHECK_LAUNCH();
    }
    return 0;
}
// row_loss: xent_scratch_floats(rows) floats of scratch (the per-row terms, then the partials of a two-stage sum); *loss_dev = their
// fixed-order sum
int64_t xent_scratch_floats(int64_t rows) { return rows < 0 ? -1 : rows + sum_partials(rows); }
int xent_rows(const float* logits, const int64_t* targets, int64_t target_stride, int32_t T, const float* weight, const float* norm_dev,
              float* loss_dev, float* row_loss, int64_t rows, const int32_t* row_pos, int32_t V, int64_t ld, void* dlogits, int32_t dl_dtype,
              int64_t ld_dl, hipStream_t s) {
    if (!logits || !targets || !weight || !norm_dev || !loss_dev || !row_loss || !dlogits || rows < 0 || V < 1 || ld < V || ld_dl < V || T < 1)
        return ORTK_EINVAL;
    if (dl_dtype != ORTK_F32 && dl_dtype != ORTK_BF16) return ORTK_EINVAL;
    if (rows == 0) return ortk_fill(loss_dev, 1, 0.f, (ortk_stream)s);
    if (V <= 256 * 40 && ld_dl <= 256 * 40 && V > 256 * 8 && dl_dtype == ORTK_BF16)
        hipLaunchKernelGGL((xent_reg_kernel<40, true>), dim3((unsigned)rows), dim3(256), 0, s, logits, targets, target_stride, T,
                           weight, norm_dev, row_loss, V, ld, dlogits, (int)dl_dtype, ld_dl, row_pos);
    else if (V <= 256 * 40 && ld_dl <= 256 * 40 && V > 256 * 8)
        hipLaunchKernelGGL((xent_reg_kernel<40, false>), dim3((unsigned)rows), dim3(256), 0, s, logits, targets, target_stride, T,
                           weight, norm_dev, row_loss, V, ld, dlogits, (int)dl_dtype, ld_dl, row_pos);
    else
        hipLaunchKernelGGL(xent_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, targets, target_stride, T, weight,
                           norm_dev, row_loss, V, ld, dlogits, (int)dl_dtype, ld_dl, row_pos);
    ORTK_CHECK_LAUNCH();
    return sum_fixed(row_loss, rows, row_loss + rows, loss_dev, s);
}
}  // namespace ortk

extern "C" int64_t ortk_xent_scratch_floats(int64_t rows) { return ortk::xent_scratch_floats(rows); }
extern "C" int ortk_xent_fwd_bwd(const float* logits, const int64_t* targets, int64_t target_stride, int32_t T, const float* weight,
                                 const float* norm_dev, float* loss_dev, float* row_loss, int64_t rows, int32_t V, int64_t ld, void* dlogits,
                                 int32_t dl_dtype, int64_t ld_dl, ortk_stream stream) {
    return ortk::xent_rows(logits, targets, target_stride, T, weight, norm_dev, loss_dev, row_loss, rows, nullptr, V, ld, dlogits, dl_dtype,
                           ld_dl, ortk_s(stream));
}

extern "C" int ortk_log_softmax_bwd(const float* logp, const float* dlogp, int64_t ld_in, void* dlogits, int32_t dl_dtype,
                                    int64_t ld_out, int64_t rows, int32_t V, ortk_stream stream) {
    if (!logp || !dlogp || !dlogits || rows < 0 || V < 1 || ld_in < V || ld_out < V) return ORTK_EINVAL;
    if (dl_dtype != ORTK_F32 && dl_dtype != ORTK_BF16) return ORTK_EINVAL;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, ortk_s(stream), logp, dlogp, ld_in, dlogits,
                       (int)dl_dtype, ld_out, V);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_colsum(const void* x, int32_t x_dtype, int64_t ld, float* out, int64_t M, int32_t N, ortk_stream stream) {
    if (!x || !out || M < 0 || N < 0 || (x_dtype != ORTK_F32 && x_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (M == 0 || N == 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ortk_cdiv(N, 64), (unsigned)ortk_cdiv(M, CS_ROWS)), dim3(256), 0,
                       ortk_s(stream), x, (int)x_dtype, ld, out, M, N);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_dropout_apply(const float* x, void* y, int32_t y_dtype, int64_t n, float p, uint32_t seed, ortk_stream stream) {
    if (!x || !y || n < 0 || p < 0.f || p >= 1.f || (y_dtype != ORTK_F32 && y_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), x, y, (int)y_dtype, n, p, seed);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_dropout_apply_rows(const float* x, void* y, int32_t y_dtype, int64_t rows, int32_t d, float p, uint32_t seed,
                                       const int32_t* drop_rows, ortk_stream stream) {
    if (rows < 0 || d < 1) return ORTK_EINVAL;
    if (!drop_rows) return ortk_dropout_apply(x, y, y_dtype, rows * d, p, seed, stream);
    if (!x || !y || p < 0.f || p >= 1.f || (y_dtype != ORTK_F32 && y_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(dropout_apply_rows_kernel, dim3(ew_grid(rows * d)), dim3(256), 0, ortk_s(stream), x, y, (int)y_dtype, rows, (int)d, p, seed, drop_rows);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_gate_apply(const float* x, const float* gate, void* y, int32_t y_dtype, int64_t n, float scale, ortk_stream stream) {
    if (!x || !gate || !y || n < 0 || (y_dtype != ORTK_F32 && y_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(gate_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), x, gate, y, (int)y_dtype, n, scale);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_cast_bf16(const float* x, void* y, int64_t n, ortk_stream stream) {
    if (!x || !y || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7)) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ortk_s(stream), x, reinterpret_cast<__bf16*>(y), n);
    ORTK_CHECK_LAUNCH();
    return 0;
}

namespace ortk {
int cast_bf16_transposed(const float* x, void* yt, const WBlockTable& t, hipStream_t s) {
    if (!x || !yt || t.n < 0 || t.n > MAX_WBLOCKS) return ORTK_EINVAL;
    if (t.n == 0 || t.tiles == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((unsigned)t.tiles), dim3(256), 0, s, x, reinterpret_cast<__bf16*>(yt), t);
    ORTK_CHECK_LAUNCH();
    return 0;
}
}  // namespace ortk

// y[r, 0..cols) += x[r, 0..cols) for two column blocks of one (rows, ld) matrix (dtype 0 fp32 / 1 bf16)
extern "C" int ortk_axpy_cols(const void* x, void* y, int32_t dtype, int64_t ld, int64_t rows, int32_t cols, ortk_stream stream) {
    if (!x || !y || rows < 0 || cols < 0 || ld < cols || (dtype != ORTK_F32 && dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(add_cols_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, ortk_s(stream), x, y, (int)dtype, ld, rows, cols);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_valid_position_tables(const int64_t* counts, int32_t R, int32_t T, int32_t* cap_off, int32_t* row_pos, ortk_stream stream) {
    if (!counts || !cap_off || !row_pos || R < 1 || R > 1024 * 64 || T < 1) return ORTK_EINVAL;
    hipLaunchKernelGGL(valid_rows_scan_kernel, dim3(1), dim3(1024), 0, ortk_s(stream), counts, (int)R, cap_off);
    ORTK_CHECK_LAUNCH();
    hipLaunchKernelGGL(valid_rows_fill_kernel, dim3((unsigned)ortk_cdiv(R, 4)), dim3(256), 0, ortk_s(stream), cap_off, (int)R, (int)T, row_pos);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_fill(float* x, int64_t n, float value, ortk_stream stream) {
    if (!x || n < 0) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), x, n, value);
    ORTK_CHECK_LAUNCH();
    return 0;
}

// *out_dev = sum(x) in a fixed order.  n <= 65 536 (every caller of the path: the (rows, T) normaliser mask): one launch, no scratch;
// longer inputs need `scratch` = ortk_sum_scratch_floats(n) floats.
extern "C" int64_t ortk_sum_scratch_floats(int64_t n) { return n < 0 ? -1 : ortk::sum_partials(n); }
extern "C" int ortk_sum(const float* x, int64_t n, float* scratch, float* out_dev, ortk_stream stream) {
    return ortk::sum_fixed(x, n, scratch, out_dev, ortk_s(stream));
}
