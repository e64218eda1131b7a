// ortk_decode.hip — on-device bookkeeping of greedy / multinomial / beam-search decoding.
//
// Replaces the host loops of CaptionModel.batch_beam_search (models/caption_model.py:56-111,151-226) and of
// CachedTransformerBase._generate_captions (models/transformer.py:507-561).  The reference does, per step, a full
// sort of b*V candidates per image, a Python loop over images x beams with two .item() syncs each, a torch.cat of
// the (N,b,t,V) log-prob history and an index_select of all 24 cache tensors.  Here:
//   * one workgroup per image selects the b best of b*V candidates (per-thread top lists -> LDS -> b rounds of
//     block arg-max), in descending order like the sort;
//   * histories are (N*b, L) token / token-log-prob tables re-ordered by parent (ping-pong buffers);
//   * the self-attention KV cache is NEVER moved: each beam carries a table of the physical cache rows of its
//     ancestors (kvidx), re-threaded by parent pointer each step; cross-attention K/V exist once per image;
//   * finished hypotheses go to a per-image list; the final stable top-b by (length-penalised) score runs on
//     device too.  No host synchronisation anywhere in the 18 steps.
#include "ortk_internal.h"

namespace ortk {
namespace {

__global__ void kv_append_kernel(const float* __restrict__ qkv, void* __restrict__ ck, void* __restrict__ cv, int kvdt, int64_t rows,
                                 int d, int row_mult, int tmax, int t) {
    const int64_t n = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        const int64_t dst = ((r * row_mult) * tmax + t) * d + c;
        st_elem(ck, dst, kvdt, qkv[r * 3 * d + d + c]);
        st_elem(cv, dst, kvdt, qkv[r * 3 * d + 2 * d + c]);
    }
}

__global__ void fill_i64_kernel(int64_t* p, int64_t n, int64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
__global__ void decode_poison_kernel(const int32_t* __restrict__ status, int64_t* __restrict__ seq, float* __restrict__ lp, float* __restrict__ score,
                                     int64_t nseq, int64_t nscore) {
    if (*status == 0) return;
    const float nan = __builtin_nanf("");
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nseq; i += (int64_t)gridDim.x * 256) { seq[i] = 0; lp[i] = nan; }
    if (score) for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nscore; i += (int64_t)gridDim.x * 256) score[i] = nan;
}
__global__ void kvidx_init_kernel(int32_t* p, int64_t rows, int row_mult, int tmax) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256)
        p[i] = (int32_t)(i * row_mult * tmax);
}

// ------------------------------------------------------------------------------------------------ beam step
constexpr int MAXB = 8;   // beams kept per thread (>= beam size)

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ double length_pen(int kind, double alpha, int len, double p) {
    if (kind == 1) return p / (pow(5.0 + (double)len, alpha) / pow(6.0, alpha));   // utils/model_utils.py:134-140
    if (kind == 2) return p / (double)len;                                         // utils/model_utils.py:143-146
    return p;
}

// same reductions, in the same order, as log_softmax_kernel (ortk_embed_loss.hip): the fused step is bit-identical to
// log_softmax followed by the unfused step
__device__ __forceinline__ float blk_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < 4; ++w) r = fmaxf(r, sh[w]);
    return r;
}
__device__ __forceinline__ float blk_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < 4; ++w) r += sh[w];
    return r;
}

// FUSED: `logp` holds raw generator logits; the row's log-soft-max (of logits * scale) is taken here — max, sum and
// candidate scan are three strided passes over the row (the 2nd and 3rd hit L2), with 8 independent loads in flight
// per thread.  Saves the separate log-soft-max launch (read + write of rows x V fp32 per step).
// FASTEXP (mixed precision only): v_exp_f32 (__expf, ~1 ulp) for the soft-max sum instead of libm's expf — the step is bound by
// the ~20 vector instructions per element of the exact function, not by its one read of the logits (111 -> 60 us per step of
// the 1 024-image beam-5 decode); the fp32 parity mode keeps the exact function (token-exact goldens).
template <bool FAST> __device__ __forceinline__ float exp_sel(float x) { return FAST ? __expf(x) : expf(x); }
// STATS (with FUSED, scale = 1): the generator GEMM has left {max, sum exp(. - max)} of every block of 64 logits (st.gstats).
// The row's log-sum-exp comes from those 2 x 158 floats; the (b + 1)-th largest block maximum is a lower bound of the row's
// b-th best admissible element (block maxima are elements of distinct blocks, at most one of them is the token the decoding
// constraint excludes), so only the b + 1 blocks at or above it are read: 1.6 KB of a 40-KB logit row.
template <bool FUSED, int NPT, bool FASTEXP = false, bool STATS = false>
__global__ __launch_bounds__(256) void beam_step_kernel(BeamState st, const float* __restrict__ logp, int t, float scale) {
    __shared__ float sv[256 * MAXB];
    __shared__ float sh_red[4];
    __shared__ float row_mx[MAXB], row_lse[MAXB];
    __shared__ int si[256 * MAXB];
    __shared__ float red_v[4];
    __shared__ int red_i[4], red_pos[4];
    __shared__ float win_v[MAXB];
    __shared__ int win_i[MAXB];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = st.b, V = st.V, L = st.L;
    const int nq = t == 0 ? 1 : b;
    const int cur = t & 1, nxt = cur ^ 1;

    float bv[MAXB];
    int bi[MAXB];
#pragma unroll
    for (int k = 0; k < MAXB; ++k) { bv[k] = -INFINITY; bi[k] = 0x7FFFFFFF; }
    bool staged = false;     // true once sv / si hold the candidates (fast path); block-uniform
    if (STATS) {
        // one WAVE per row (rows q = wave, wave + 4): the 158 block statistics sit 3 per lane, every reduction is a wave
        // reduction and the b + 1 block reads of a row are issued together — the rows of an image proceed in parallel instead
        // of as a chain of block-wide barriers and dependent loads
        __shared__ int cand_cnt2;
        for (int k = 0; k < MAXB; ++k) { sv[tid * MAXB + k] = -INFINITY; si[tid * MAXB + k] = 0x7FFFFFFF; }
        if (tid == 0) cand_cnt2 = 0;
        __syncthreads();
        const int nblk = st.nblk;                     // <= 256: at most 4 blocks per lane
        const int want = min(b + 1, nblk);            // blocks to read per row
        for (int q = wave; q < nq; q += 4) {
            const int64_t srow = t == 0 ? (int64_t)img : (int64_t)img * b + q;
            const float* gs = st.gstats + srow * nblk * 2;
            float m[4], sb_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int blk = lane + 64 * u;
                const float2 v2 = blk < nblk ? *reinterpret_cast<const float2*>(gs + 2 * blk) : make_float2(-INFINITY, 0.f);
                m[u] = v2.x; sb_[u] = v2.y;
            }
            const float cum = t == 0 ? 0.f : st.cum[(int64_t)img * b + q];
            const int prev = (st.decoding_constraint && t > 0) ? st.seq[cur][((int64_t)img * b + q) * L + t - 1] : -1;
            const float mx = wave_max(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])));
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += m[u] > -INFINITY ? sb_[u] * __expf(m[u] - mx) : 0.f;
            const float lse = logf(wave_sum(sum));
            if (lane == 0) { row_mx[q] = mx; row_lse[q] = lse; }
            // the `want` largest block maxima, one wave arg-max round each (ties: the lower block)
            int selb[MAXB + 1];
            float tau_raw = -INFINITY;
#pragma unroll
            for (int r = 0; r < MAXB + 1; ++r) {
                selb[r] = -1;
                if (r < want) {
                    float bv_ = m[0]; int bb = lane;
#pragma unroll
                    for (int u = 1; u < 4; ++u) if (m[u] > bv_) { bv_ = m[u]; bb = lane + 64 * u; }
#pragma unroll
                    for (int o2 = 32; o2 > 0; o2 >>= 1) {
                        const float ov = __shfl_xor(bv_, o2, 64); const int ob = __shfl_xor(bb, o2, 64);
                        if (ov > bv_ || (ov == bv_ && ob < bb)) { bv_ = ov; bb = ob; }
                    }
                    selb[r] = bb; tau_raw = bv_;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (bb == lane + 64 * u) m[u] = -INFINITY;      // (never selected twice)
                }
            }
            const float tau = cum + ((tau_raw - mx) - lse);
            float zv[MAXB + 1];
#pragma unroll
            for (int r = 0; r < MAXB + 1; ++r) {
                const int v = selb[r] * 64 + lane;
                zv[r] = (selb[r] >= 0 && v < V) ? logp[srow * st.ldv + v] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < MAXB + 1; ++r) {
                const int v = selb[r] * 64 + lane;
                if (selb[r] >= 0 && v < V && v != prev) {
                    const float z = cum + ((zv[r] - mx) - lse);
                    if (z >= tau) {
                        const int pos = atomicAdd(&cand_cnt2, 1);
                        if (pos < 256 * MAXB) { sv[pos] = z; si[pos] = q * V + v; }
                    }
                }
            }
        }
        __syncthreads();
        staged = cand_cnt2 <= 256 * MAXB;          // (massive ties: the exact per-thread lists below)
        __syncthreads();
    } else if (FUSED && NPT > 0) {
        // Register-resident rows: each thread holds its NPT strided elements of a row (ONE memory round trip per row, the
        // next row's loads are issued before the current row is reduced); max and sum-exp run in the element order of
        // log_softmax_kernel.  Candidate selection without per-thread sorted lists (their insertion network ran for almost
        // every element because some lane of the wave always inserted): per row, tau = the b-th largest of the 16
        // half-wave-quarter maxima is a lower bound of the row's b-th best, so only the handful of elements >= tau go to
        // the LDS candidate list; the b block-wide arg-max rounds below then pick the exact top-b (value, then index).
        __shared__ float gmax[16];
        __shared__ int cand_cnt;
        for (int k = 0; k < MAXB; ++k) { sv[tid * MAXB + k] = -INFINITY; si[tid * MAXB + k] = 0x7FFFFFFF; }
        if (tid == 0) cand_cnt = 0;
        float zn[NPT > 0 ? NPT : 1];
        {
            const float* lp0 = logp + (t == 0 ? (int64_t)img : (int64_t)img * b) * st.ldv;
#pragma unroll
            for (int u = 0; u < NPT; ++u) { const int v = tid + 256 * u; zn[u] = v < V ? lp0[v] : 0.f; }
        }
        for (int q = 0; q < nq; ++q) {
            float z[NPT > 0 ? NPT : 1];
#pragma unroll
            for (int u = 0; u < NPT; ++u) z[u] = zn[u];
            if (q + 1 < nq) {
                const float* lpn = logp + ((int64_t)img * b + q + 1) * st.ldv;
#pragma unroll
                for (int u = 0; u < NPT; ++u) { const int v = tid + 256 * u; zn[u] = v < V ? lpn[v] : 0.f; }
            }
            const float cum = t == 0 ? 0.f : st.cum[(int64_t)img * b + q];
            const int prev = (st.decoding_constraint && t > 0) ? st.seq[cur][((int64_t)img * b + q) * L + t - 1] : -1;
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < NPT; ++u) if (tid + 256 * u < V) m = fmaxf(m, z[u] * scale);
            const float mx = blk_max(m, sh_red);
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < NPT; ++u) if (tid + 256 * u < V) sum += exp_sel<FASTEXP>(z[u] * scale - mx);
            sum = blk_sum(sum, sh_red);
            const float lse = logf(sum);
            if (tid == 0) { row_mx[q] = mx; row_lse[q] = lse; }
            float tm = -INFINITY;
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int v = tid + 256 * u;
                const bool ok = v < V && v != prev;
                z[u] = ok ? cum + ((z[u] * scale - mx) - lse) : -INFINITY;
                tm = fmaxf(tm, z[u]);
            }
#pragma unroll
            for (int o2 = 8; o2 > 0; o2 >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o2, 64));
            __syncthreads();                       // gmax of the previous row fully consumed
            if ((lane & 15) == 0) gmax[wave * 4 + (lane >> 4)] = tm;
            __syncthreads();
            float gm[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) gm[u] = gmax[u];
            float tau = -INFINITY;
            for (int r = 0; r < b; ++r) {
                float best = gm[0];
#pragma unroll
                for (int u = 1; u < 16; ++u) best = fmaxf(best, gm[u]);
                tau = best;
                bool removed = false;
#pragma unroll
                for (int u = 0; u < 16; ++u) if (!removed && gm[u] == best) { gm[u] = -INFINITY; removed = true; }
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int v = tid + 256 * u;
                if (v < V && v != prev && z[u] >= tau) {
                    const int pos = atomicAdd(&cand_cnt, 1);
                    if (pos < 256 * MAXB) { sv[pos] = z[u]; si[pos] = q * V + v; }
                }
            }
        }
        __syncthreads();
        staged = cand_cnt <= 256 * MAXB;           // overflow (massive ties): redo with the exact per-thread lists
        __syncthreads();
    }
    if (!staged) {
    for (int q = 0; q < nq; ++q) {
        const int64_t srow = t == 0 ? img : (int64_t)img * b + q;
        const float* lp = logp + srow * st.ldv;
        const float cum = t == 0 ? 0.f : st.cum[(int64_t)img * b + q];
        const int prev = (st.decoding_constraint && t > 0) ? st.seq[cur][((int64_t)img * b + q) * L + t - 1] : -1;
        float mx = 0.f, lse = 0.f;
        if (FUSED) {
            float m = -INFINITY;
            for (int v0 = tid; v0 < V; v0 += 256 * 8) {
                float z[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int v = v0 + 256 * u; z[u] = v < V ? lp[v] * scale : -INFINITY; }
#pragma unroll
                for (int u = 0; u < 8; ++u) m = fmaxf(m, z[u]);
            }
            mx = blk_max(m, sh_red);
            float sum = 0.f;
            for (int v0 = tid; v0 < V; v0 += 256 * 8) {
                float z[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int v = v0 + 256 * u; z[u] = v < V ? lp[v] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (v0 + 256 * u < V) sum += expf(z[u] * scale - mx);
            }
            sum = blk_sum(sum, sh_red);
            lse = logf(sum);
            if (tid == 0) { row_mx[q] = mx; row_lse[q] = lse; }
        }
        for (int v0 = tid; v0 < V; v0 += 256 * 8) {
            float z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int v = v0 + 256 * u; z[u] = v < V ? lp[v] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + 256 * u;
                if (v >= V || v == prev) continue;
                float cv = cum + (FUSED ? (z[u] * scale - mx) - lse : z[u]);
                int ci = q * V + v;
                if (better(cv, ci, bv[MAXB - 1], bi[MAXB - 1])) {
#pragma unroll
                    for (int k = 0; k < MAXB; ++k) {
                        if (better(cv, ci, bv[k], bi[k])) {
                            const float tv = bv[k]; const int ti = bi[k];
                            bv[k] = cv; bi[k] = ci; cv = tv; ci = ti;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) { sv[tid * MAXB + k] = bv[k]; si[tid * MAXB + k] = bi[k]; }
    }
    __syncthreads();
    // b rounds of block-wide arg-max over the 256*MAXB staged candidates (each thread scans its own MAXB)
    for (int r = 0; r < b; ++r) {
        float mv = -INFINITY; int mi = 0x7FFFFFFF, mp = -1;
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
            const float v = sv[tid * MAXB + k]; const int i = si[tid * MAXB + k];
            if (better(v, i, mv, mi)) { mv = v; mi = i; mp = tid * MAXB + k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mv, o, 64); const int oi = __shfl_xor(mi, o, 64); const int op = __shfl_xor(mp, o, 64);
            if (better(ov, oi, mv, mi)) { mv = ov; mi = oi; mp = op; }
        }
        if (lane == 0) { red_v[wave] = mv; red_i[wave] = mi; red_pos[wave] = mp; }
        __syncthreads();
        if (tid == 0) {
            float fv = red_v[0]; int fi = red_i[0], fp = red_pos[0];
            for (int w = 1; w < 4; ++w)
                if (better(red_v[w], red_i[w], fv, fi)) { fv = red_v[w]; fi = red_i[w]; fp = red_pos[w]; }
            win_v[r] = fv; win_i[r] = fi;
            if (fp >= 0) sv[fp] = -INFINITY, si[fp] = 0x7FFFFFFF;
        }
        __syncthreads();
    }
    // histories, ancestry table, next tokens.  Every copy is spread over the whole workgroup: one thread per beam walking
    // its t history entries and t+1 ancestry entries was a chain of ~50 dependent global round trips (the tail of this kernel
    // grew with t and dominated it: 118 us per step at 1 024 images x 5 beams).
    const int64_t row0 = (int64_t)img * b;
    for (int idx = tid; idx < b * t; idx += 256) {
        const int q = idx / t, u = idx - q * t;
        const int parent = win_i[q] / V;
        const int64_t nrow = row0 + q, prow = row0 + parent;
        st.seq[nxt][nrow * L + u] = st.seq[cur][prow * L + u];
        st.tok_lp[nxt][nrow * L + u] = st.tok_lp[cur][prow * L + u];
    }
    for (int idx = tid; idx < b * (t + 1); idx += 256) {
        const int q = idx / (t + 1), u = idx - q * (t + 1);
        const int parent = win_i[q] / V;
        const int64_t srow = t == 0 ? img : row0 + parent;
        const int32_t v = st.kvidx[cur][srow * (t + 1) + u];
        st.kvidx[nxt][(row0 + q) * (t + 2) + u] = v;                                   // ancestors' cache rows
        if (st.uniq) {      // measurement only: is this the first of the image's new beams that references cache row v?
            bool first = true;
            for (int q2 = 0; q2 < q; ++q2) {
                const int64_t s2 = t == 0 ? img : row0 + win_i[q2] / V;
                if (st.kvidx[cur][s2 * (t + 1) + u] == v) first = false;
            }
            if (first) atomicAdd(st.uniq, 1ull);
        }
    }
    __shared__ int end_slot[MAXB];
    if (tid < b) {
        const int q = tid;
        const int ix = win_i[q];
        const int parent = ix / V, tok = ix - parent * V;
        const int64_t nrow = row0 + q, prow = row0 + parent;
        const int64_t srow = t == 0 ? img : prow;
        st.seq[nxt][nrow * L + t] = tok;
        const int pq = t == 0 ? 0 : parent;
        st.tok_lp[nxt][nrow * L + t] = FUSED ? (logp[srow * st.ldv + tok] * scale - row_mx[pq]) - row_lse[pq] : logp[srow * st.ldv + tok];
        st.it[nrow] = tok;
        st.kvidx[nxt][nrow * (t + 2) + t + 1] = (int32_t)(nrow * st.tmax + t + 1);      // this beam's own slot at time t+1
    }
    // finished hypotheses: slots are handed out sequentially per image (insertion order matters for the final stable
    // sort), the copies run in parallel afterwards
    const int cap = b * L;
    if (tid == 0) {
        int cnt = st.done_cnt[img];
        for (int q = 0; q < b; ++q) {
            const int64_t nrow = row0 + q;
            const int ix = win_i[q];
            const int tok = ix - (ix / V) * V;
            float cum = win_v[q];
            const bool end = tok == st.eos || t == L - 1;
            end_slot[q] = -1;
            if (end) {
                if (cnt < cap) {
                    const int64_t base = ((int64_t)img * cap + cnt);
                    end_slot[q] = cnt;
                    st.done_len[base] = t + 1;
                    st.done_p[base] = length_pen(st.length_penalty, st.length_alpha, t + 1, (double)cum);
                    ++cnt;
                }
                cum -= 1000.f;
            }
            st.cum[nrow] = cum;
        }
        st.done_cnt[img] = cnt;
    }
    __syncthreads();          // end_slot, and the histories written above, are visible to the whole workgroup
    for (int idx = tid; idx < b * (t + 1); idx += 256) {
        const int q = idx / (t + 1), u = idx - q * (t + 1);
        if (end_slot[q] < 0) continue;
        const int64_t base = (int64_t)img * cap + end_slot[q], nrow = row0 + q;
        st.done_seq[base * L + u] = st.seq[nxt][nrow * L + u];
        st.done_lp[base * L + u] = st.tok_lp[nxt][nrow * L + u];
    }
}

// stable top-b of each image's finished list by score (descending; earlier insertion wins ties)
__global__ __launch_bounds__(64) void beam_finalize_kernel(BeamState st, int64_t* __restrict__ seq_out, float* __restrict__ lp_out,
                                                           float* __restrict__ score_out) {
    __shared__ unsigned char taken[MAXB * 64];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int b = st.b, L = st.L, cap = b * L;
    const int cnt = st.done_cnt[img];
    for (int i = lane; i < cap; i += 64) taken[i] = 0;
    __syncthreads();
    for (int r = 0; r < b; ++r) {
        double best = -INFINITY; int bi = 0x7FFFFFFF;
        for (int i = lane; i < cnt; i += 64)
            if (!taken[i]) {
                const double p = st.done_p[(int64_t)img * cap + i];
                if (p > best || (p == best && i < bi)) { best = p; bi = i; }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double op = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (op > best || (op == best && oi < bi)) { best = op; bi = oi; }
        }
        const int64_t orow = (int64_t)img * b + r;
        if (bi != 0x7FFFFFFF) {
            const int64_t base = (int64_t)img * cap + bi;
            const int len = st.done_len[base];
            for (int u = lane; u < L; u += 64) {
                seq_out[orow * L + u] = u < len ? (int64_t)st.done_seq[base * L + u] : 0;
                lp_out[orow * L + u] = u < len ? st.done_lp[base * L + u] : 0.f;
            }
            if (lane == 0) { taken[bi] = 1; if (score_out) score_out[orow] = (float)best; }
        } else {
            for (int u = lane; u < L; u += 64) { seq_out[orow * L + u] = 0; lp_out[orow * L + u] = 0.f; }
            if (lane == 0 && score_out) score_out[orow] = 0.f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ greedy / multinomial
__global__ void sample_init_kernel(SampleState st, int bos) {
    const int64_t n = (int64_t)st.rows * st.L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { st.seq[i] = 0; st.lp[i] = 0.f; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < st.rows; i += (int64_t)gridDim.x * 256) {
        st.it[i] = bos; st.unfinished[i] = bos != st.eos;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st.last_step[0] = -1; st.last_step[1] = -1; }
}

// FAST (mixed precision only, with the fast exponential of the soft-max): v_log_f32 for the two logarithms — the same uniforms,
// the noise to ~1 ulp (the SCST step did not move measurably: 23.2 ms either way)
template <bool FAST = false>
__device__ __forceinline__ float gumbel(uint64_t seed, int t, int row, int v) { return ortk_gumbel<FAST>(seed, t, row, v); }

__global__ __launch_bounds__(256) void sample_step_kernel(SampleState st, const float* __restrict__ logp, int t) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lp = logp + (int64_t)row * st.ldv;
    const int prev = (st.decoding_constraint && t > 0) ? (int)st.seq[(int64_t)row * st.L + t - 1] : -1;
    const bool is_greedy = st.greedy_stride > 0 && row % st.greedy_stride == 0;
    const bool samp = st.sample && !is_greedy;
    // the hash is keyed by the row index the sample would have in a samples-only call
    const int64_t grow = st.row_offset + row;
    const int hrow = (int)(st.greedy_stride > 0 ? grow - grow / st.greedy_stride - 1 : grow);
    float mv = -INFINITY; int mi = 0x7FFFFFFF;
    for (int v = tid; v < st.V; v += 256) {
        if (v == prev) continue;
        float x = lp[v];
        if (samp) x = x / st.temperature + gumbel(st.seed, t, hrow, v);
        if (better(x, v, mv, mi)) { mv = x; mi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mv, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (better(ov, oi, mv, mi)) { mv = ov; mi = oi; }
    }
    if (lane == 0) { red_v[wave] = mv; red_i[wave] = mi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w], red_i[w], mv, mi)) { mv = red_v[w]; mi = red_i[w]; }
        const int unf = st.unfinished[row];
        st.it[row] = mi;
        st.seq[(int64_t)row * st.L + t] = unf ? mi : 0;      // seq[:, t] = it * unfinished   (transformer.py:546)
        st.lp[(int64_t)row * st.L + t] = lp[mi];             // NOT masked after EOS          (transformer.py:548)
        const int now = unf && (mi != st.eos);
        st.unfinished[row] = now;
        // the reference leaves the loop at the first step where no row is unfinished (transformer.py:550-551)
        int32_t* last = st.last_step + (is_greedy ? 1 : 0);
        if (unf && !now) atomicMax(last, t);
        if (now && t == st.L - 1) atomicMax(last, t);
    }
}

// Same step on RAW generator logits: the row stays in registers (one read), its log-soft-max statistics are taken with the
// reductions of log_softmax_kernel (bit-identical log-probabilities), then the arg-max / Gumbel-max runs on
// logp = (logit - max) - lse.  Saves the separate log-soft-max pass (read + write of rows x V fp32 per step).
template <int NPT, bool FASTEXP = false>
__global__ __launch_bounds__(256) void sample_step_fused_kernel(SampleState st, const float* __restrict__ logits, int t) {
    __shared__ float sh_red[4];
    __shared__ float red_v[4], red_l[4];
    __shared__ int red_i[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lp = logits + (int64_t)row * st.ldv;
    float z[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) { const int v = tid + 256 * u; z[u] = v < st.V ? lp[v] : 0.f; }
    const int prev = (st.decoding_constraint && t > 0) ? (int)st.seq[(int64_t)row * st.L + t - 1] : -1;
    const bool is_greedy = st.greedy_stride > 0 && row % st.greedy_stride == 0;
    const bool samp = st.sample && !is_greedy;
    const int64_t grow = st.row_offset + row;
    const int hrow = (int)(st.greedy_stride > 0 ? grow - grow / st.greedy_stride - 1 : grow);
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < NPT; ++u) if (tid + 256 * u < st.V) m = fmaxf(m, z[u]);
    const float mx = blk_max(m, sh_red);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NPT; ++u) if (tid + 256 * u < st.V) sum += exp_sel<FASTEXP>(z[u] - mx);
    sum = blk_sum(sum, sh_red);
    const float lse = logf(sum);
    float mv = -INFINITY, ml = 0.f; int mi = 0x7FFFFFFF;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int v = tid + 256 * u;
        if (v >= st.V || v == prev) continue;
        const float l = (z[u] - mx) - lse;
        float x = l;
        if (samp) x = x / st.temperature + gumbel<FASTEXP>(st.seed, t, hrow, v);
        if (better(x, v, mv, mi)) { mv = x; mi = v; ml = l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mv, o, 64), ol = __shfl_xor(ml, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (better(ov, oi, mv, mi)) { mv = ov; mi = oi; ml = ol; }
    }
    if (lane == 0) { red_v[wave] = mv; red_i[wave] = mi; red_l[wave] = ml; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w], red_i[w], mv, mi)) { mv = red_v[w]; mi = red_i[w]; ml = red_l[w]; }
        const int unf = st.unfinished[row];
        st.it[row] = mi;
        st.seq[(int64_t)row * st.L + t] = unf ? mi : 0;
        st.lp[(int64_t)row * st.L + t] = ml;
        const int now = unf && (mi != st.eos);
        st.unfinished[row] = now;
        int32_t* last = st.last_step + (is_greedy ? 1 : 0);
        if (unf && !now) atomicMax(last, t);
        if (now && t == st.L - 1) atomicMax(last, t);
    }
}

// The sampling step on the generator's own epilogue output (ortk_gemm_args.tile_samp + tile_stats): per row and block of 64 logits the
// best Gumbel-max candidate {key, column, logit} and the soft-max partials {max, sum exp}.  One wave per row: arg-max over the blocks
// (same total order as the row kernels: larger key, lower column on a tie), log-sum-exp from the partials, log-prob of the chosen token,
// then the bookkeeping of sample_step_fused_kernel.  The logit rows (62 MB per position of the SCST rollout) are neither written nor read.
template <bool FASTEXP>
__global__ __launch_bounds__(256) void sample_combine_kernel(SampleState st, const float* __restrict__ gstats, const float* __restrict__ gsamp, int nblk, int t) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= st.rows) return;
    const float* gs = gstats + (int64_t)row * nblk * 2;
    const float* gp = gsamp + (int64_t)row * nblk * 4;
    float m = -INFINITY, mv = -INFINITY, mz = 0.f; int mi = 0x7FFFFFFF;
    for (int b = lane; b < nblk; b += 64) {
        m = fmaxf(m, gs[2 * b]);
        const float x = gp[4 * b]; const int i = __float_as_int(gp[4 * b + 1]);
        if (ortk_better(x, i, mv, mi)) { mv = x; mi = i; mz = gp[4 * b + 2]; }
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int b = lane; b < nblk; b += 64) { const float sb = gs[2 * b + 1]; if (sb > 0.f) sum += sb * exp_sel<FASTEXP>(gs[2 * b] - m); }
    sum = wave_sum(sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mv, o, 64), oz = __shfl_xor(mz, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (ortk_better(ov, oi, mv, mi)) { mv = ov; mi = oi; mz = oz; }
    }
    if (lane == 0) {
        const float ml = (mz - m) - logf(sum);
        const bool is_greedy = st.greedy_stride > 0 && row % st.greedy_stride == 0;
        const int unf = st.unfinished[row];
        st.it[row] = mi;
        st.seq[(int64_t)row * st.L + t] = unf ? mi : 0;
        st.lp[(int64_t)row * st.L + t] = ml;
        const int now = unf && (mi != st.eos);
        st.unfinished[row] = now;
        int32_t* last = st.last_step + (is_greedy ? 1 : 0);
        if (unf && !now) atomicMax(last, t);
        if (now && t == st.L - 1) atomicMax(last, t);
    }
}

__global__ void sample_finalize_kernel(SampleState st) {
    const int64_t n = (int64_t)st.rows * st.L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % st.L);
        const int64_t row = i / st.L;
        const int last = st.last_step[(st.greedy_stride > 0 && row % st.greedy_stride == 0) ? 1 : 0];
        if (t > last) { st.lp[i] = 0.f; st.seq[i] = 0; }
    }
}

inline unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ortk_cdiv(n, 256), 2048)); }

}  // namespace

int kv_append(const float* qkv, void* ck, void* cv, int32_t kvdt, int64_t rows, int32_t d, int32_t row_mult, int32_t tmax, int32_t t,
              hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(kv_append_kernel, dim3(ew_grid(rows * d)), dim3(256), 0, s, qkv, ck, cv, (int)kvdt, rows, d, row_mult, tmax, t);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int decode_poison(const int32_t* status, int64_t* seq, float* lp, float* score, int64_t nseq, int64_t nscore, hipStream_t s) {
    hipLaunchKernelGGL(decode_poison_kernel, dim3(64), dim3(256), 0, s, status, seq, lp, score, nseq, nscore);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int fill_i64(int64_t* p, int64_t n, int64_t v, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_i64_kernel, dim3(ew_grid(n)), dim3(256), 0, s, p, n, v);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int fill_i32(int32_t* p, int64_t n, int32_t v, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(ew_grid(n)), dim3(256), 0, s, p, n, v);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int kvidx_init(int32_t* kvidx, int64_t rows, int32_t row_mult, int32_t tmax, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(kvidx_init_kernel, dim3(ew_grid(rows)), dim3(256), 0, s, kvidx, rows, row_mult, tmax);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int beam_step(const BeamState& st, const float* logp, int32_t t, hipStream_t s, bool fused, float scale, bool fast_exp) {
    if (st.b < 1 || st.b > MAXB) return ORTK_EINVAL;
    if (st.B == 0) return 0;
    if (fused && fast_exp && st.gstats && scale == 1.f && st.nblk <= 256 && st.nblk * 64 >= st.V)
        hipLaunchKernelGGL((beam_step_kernel<true, 0, true, true>), dim3((unsigned)st.B), dim3(256), 0, s, st, logp, t, scale);
    else if (fused && st.V <= 256 * 40 && fast_exp) hipLaunchKernelGGL((beam_step_kernel<true, 40, true>), dim3((unsigned)st.B), dim3(256), 0, s, st, logp, t, scale);
    else if (fused && st.V <= 256 * 40) hipLaunchKernelGGL((beam_step_kernel<true, 40>), dim3((unsigned)st.B), dim3(256), 0, s, st, logp, t, scale);
    else if (fused) hipLaunchKernelGGL((beam_step_kernel<true, 0>), dim3((unsigned)st.B), dim3(256), 0, s, st, logp, t, scale);
    else            hipLaunchKernelGGL((beam_step_kernel<false, 0>), dim3((unsigned)st.B), dim3(256), 0, s, st, logp, t, 1.f);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int beam_finalize(const BeamState& st, int64_t* seq_out, float* lp_out, float* score_out, hipStream_t s) {
    if (st.b < 1 || st.b > MAXB || st.b * st.L > MAXB * 64) return ORTK_EINVAL;
    if (st.B == 0) return 0;
    hipLaunchKernelGGL(beam_finalize_kernel, dim3((unsigned)st.B), dim3(64), 0, s, st, seq_out, lp_out, score_out);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int sample_init(const SampleState& st, int32_t bos, hipStream_t s) {
    hipLaunchKernelGGL(sample_init_kernel, dim3(ew_grid((int64_t)st.rows * st.L)), dim3(256), 0, s, st, bos);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int sample_step(const SampleState& st, const float* logp, int32_t t, hipStream_t s, bool fused, bool fast_exp) {
    if (st.rows == 0) return 0;
    if (fused) {       // `logp` holds raw logits (V <= 10 240: the caller checks sample_step_can_fuse)
        if (st.V > 256 * 40) return ORTK_EINVAL;
        if (fast_exp) hipLaunchKernelGGL((sample_step_fused_kernel<40, true>), dim3((unsigned)st.rows), dim3(256), 0, s, st, logp, t);
        else hipLaunchKernelGGL(sample_step_fused_kernel<40>, dim3((unsigned)st.rows), dim3(256), 0, s, st, logp, t);
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(sample_step_kernel, dim3((unsigned)st.rows), dim3(256), 0, s, st, logp, t);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int sample_combine(const SampleState& st, const float* gstats, const float* gsamp, int32_t nblk, int32_t t, hipStream_t s, bool fast_exp) {
    if (st.rows == 0) return 0;
    const dim3 grid((unsigned)((st.rows + 3) / 4));
    if (fast_exp) hipLaunchKernelGGL(sample_combine_kernel<true>, grid, dim3(256), 0, s, st, gstats, gsamp, (int)nblk, (int)t);
    else hipLaunchKernelGGL(sample_combine_kernel<false>, grid, dim3(256), 0, s, st, gstats, gsamp, (int)nblk, (int)t);
    ORTK_CHECK_LAUNCH();
    return 0;
}
int sample_finalize(const SampleState& st, hipStream_t s) {
    if (st.rows == 0) return 0;
    hipLaunchKernelGGL(sample_finalize_kernel, dim3(ew_grid((int64_t)st.rows * st.L)), dim3(256), 0, s, st);
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace ortk
