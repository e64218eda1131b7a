// ortk_attn.hip — scaled-dot-product attention for the small tiles of the ORT path, forward and backward.
//
// Replaces MultiHeadedAttention.attention (models/transformer.py:285-295) and
// BoxMultiHeadedAttention.box_attention (models/relation_transformer.py:258-293):
//     score = q.k / sqrt(dk);  score[mask == 0] = -1e9;  score += log-geometry bias;  P = softmax(score);
//     O = dropout(P) V
// Tiles are tiny (36x36, 17x17, 17x36 keys x queries with dk = 64), so one 256-thread workgroup owns one
// (key/value group, head): K and V of the group sit in LDS (row pitch dk+1 -> conflict-free for both the
// "lane = key" dot products and the "lane = feature" accumulations), each wave walks query rows, softmax
// statistics are wavefront shuffles.  A key/value group serves Lq consecutive query rows — for the decoder's
// cross attention that is all captions of one image (5 x 17 rows), so the encoder memory's K/V are read once
// per image instead of once per caption (the reference repeats them, relation_transformer.py:63-66), and in
// the backward dK/dV of an image are complete inside one workgroup (no atomics to global memory).
#include <cstdlib>
#include "ortk_internal.h"

namespace {

constexpr int MAXK = 128;   // keys per group (2 per lane)
constexpr int MAXD = 64;    // head dim
constexpr int KP = MAXD + 1;

// broadcast lane `l` (compile-time constant after unrolling) of a float: v_readlane_b32 -> SGPR
__device__ __forceinline__ float bcast(float v, int l) { return __shfl(v, l, 64); }

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// LDS is carved at run time from the group's actual key count (more workgroups per CU for short tiles).
struct Tiles {
    float (*k)[KP];      // [Lk][KP]
    float (*v)[KP];      // [Lk][KP]
    float* mask;         // [MAXK]
    float (*q)[MAXD];    // [4][MAXD]
    float (*p)[MAXK];    // [4][MAXK]
};
__host__ __device__ inline size_t fwd_lds_bytes(int Lk) {
    return sizeof(float) * ((size_t)2 * Lk * KP + MAXK + 4 * MAXD + 4 * MAXK);
}
__device__ __forceinline__ Tiles carve_fwd(float* base, int Lk) {
    Tiles t;
    t.k = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.v = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.mask = base; base += MAXK;
    t.q = reinterpret_cast<float (*)[MAXD]>(base); base += 4 * MAXD;
    t.p = reinterpret_cast<float (*)[MAXK]>(base);
    return t;
}

__device__ __forceinline__ void load_kv(Tiles& t, const ortk_attn_args& a, int g, int h) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < a.Lk * a.dk; idx += 256) {
        const int j = idx / a.dk, dd = idx - j * a.dk;
        const int64_t row = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * a.Lk + j]
                                       : (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : a.Lk) + j;
        t.k[j][dd] = a.k[row * a.ldk + h * a.dk + dd];
        t.v[j][dd] = a.v[row * a.ldv + h * a.dk + dd];
    }
    for (int j = tid; j < a.Lk; j += 256) t.mask[j] = a.kmask ? a.kmask[(int64_t)g * a.Lk + j] : 1.f;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(ortk_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    Tiles t = carve_fwd(smem_f, a.Lk);
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_kv(t, a, g, h);
    __syncthreads();
    const float scale = sqrtf((float)a.dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    for (int i = wave; i < a.Lq; i += 4) {
        const int64_t qrow = (int64_t)g * a.Lq + i;
        if (lane < a.dk) t.q[wave][lane] = a.q[qrow * a.ldq + h * a.dk + lane];
        wave_sync();
        const int qpos = a.causal_period > 0 ? i % a.causal_period : 0;
        const int64_t pbase = (((int64_t)g * a.H + h) * a.Lq + i) * a.Lk;
        // dropout draws: the natural index, or (decode step in train mode) the teacher-forced pass's (ortk_attn_args.drop_tf_*)
        const int64_t dbase = a.drop_tf_T > 0 ? (((int64_t)g * a.H + h) * ((int64_t)a.Lq * a.drop_tf_T) + (int64_t)i * a.drop_tf_T + a.drop_tf_t) * a.drop_tf_lk
                                              : pbase;
        float s[2], mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            s[u] = -INFINITY;
            if (j < a.Lk) {
                float acc = 0.f;
                for (int dd = 0; dd < a.dk; ++dd) acc += t.q[wave][dd] * t.k[j][dd];
                acc = acc / scale;
                const bool masked = (t.mask[j] == 0.f) || (a.causal_period > 0 && j > qpos);
                if (masked) acc = -1e9f;
                if (a.bias) acc = a.bias[pbase + j] + acc;
                s[u] = acc;
            }
            mx = fmaxf(mx, s[u]);
        }
        mx = wave_max(mx);
        float e[2], sum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) { e[u] = (lane + 64 * u < a.Lk) ? expf(s[u] - mx) : 0.f; sum += e[u]; }
        sum = wave_sum(sum);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < a.Lk) {
                float p = e[u] / sum;
                if (a.p) a.p[pbase + j] = p;
                if (a.drop_p > 0.f) p = ortk_keep(a.drop_seed, (uint64_t)(dbase + j), a.drop_p) ? p * inv_keep : 0.f;
                t.p[wave][j] = p;
            }
        }
        wave_sync();
        if (lane < a.dk) {
            float o = 0.f;
            for (int j = 0; j < a.Lk; ++j) o += t.p[wave][j] * t.v[j][lane];
            st_elem(a.o, qrow * a.ldo + h * a.dk + lane, a.o_dtype, o);
        }
        wave_sync();
    }
}

// Forward, fast form (Lk <= 64): one wave per (K/V group, head) pair, no workgroup barrier (see the backward below).
__global__ __launch_bounds__(256) void attn_fwd_wave_kernel(ortk_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nkv * a.H) return;
    const int g = pair / a.H, h = pair - g * a.H;
    const int Lk = a.Lk, dk = a.dk;
    const size_t per_wave = (size_t)2 * Lk * KP + 2 * 64;
    float* base = smem_f + wave * per_wave;
    float (*sK)[KP] = reinterpret_cast<float (*)[KP]>(base);
    float (*sV)[KP] = reinterpret_cast<float (*)[KP]>(base + (size_t)Lk * KP);
    float* sq = base + (size_t)2 * Lk * KP;
    float* sp = sq + 64;
    for (int idx = lane; idx < Lk * KP; idx += 64) {
        const int j = idx / KP, dd = idx - j * KP;
        const int64_t row = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * Lk + j]
                                       : (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : Lk) + j;
        const bool in = dd < dk;
        sK[j][dd] = in ? a.k[row * a.ldk + h * dk + dd] : 0.f;
        sV[j][dd] = in ? a.v[row * a.ldv + h * dk + dd] : 0.f;
    }
    const float kmask = (lane < Lk && a.kmask) ? a.kmask[(int64_t)g * Lk + lane] : 1.f;
    const float scale = sqrtf((float)dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int jl = lane < Lk ? lane : 0, dl = lane < dk ? lane : 0;
    wave_sync();
    for (int i = 0; i < a.Lq; ++i) {
        const int64_t qrow = (int64_t)g * a.Lq + i;
        sq[lane] = lane < dk ? a.q[qrow * a.ldq + h * dk + lane] : 0.f;
        const int qpos = a.causal_period > 0 ? i % a.causal_period : 0;
        const int64_t pbase = (((int64_t)g * a.H + h) * a.Lq + i) * Lk;
        wave_sync();
        float acc = 0.f;
#pragma unroll 8
        for (int dd = 0; dd < MAXD; ++dd) acc += sq[dd] * sK[jl][dd];
        float sc = -INFINITY;
        if (lane < Lk) {
            acc = acc / scale;
            if (kmask == 0.f || (a.causal_period > 0 && lane > qpos)) acc = -1e9f;
            if (a.bias) acc = a.bias[pbase + lane] + acc;
            sc = acc;
        }
        const float mx = wave_max(sc);
        const float e = lane < Lk ? expf(sc - mx) : 0.f;
        const float sum = wave_sum(e);
        float p = e / sum;
        if (lane < Lk) {
            if (a.p) a.p[pbase + lane] = p;
            if (a.drop_p > 0.f) p = ortk_keep(a.drop_seed, (uint64_t)(pbase + lane), a.drop_p) ? p * inv_keep : 0.f;
        }
        sp[lane] = lane < Lk ? p : 0.f;
        wave_sync();
        float o = 0.f;
        for (int j = 0; j < Lk; ++j) o += sp[j] * sV[j][dl];
        if (lane < dk) st_elem(a.o, qrow * a.ldo + h * dk + lane, a.o_dtype, o);
        wave_sync();
    }
}

// ================================================================================================ MFMA attention
// fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, so the parity mode keeps its numerics) for the four small
// products of a (K/V group, head) pair.  The per-lane-FMA kernels above are LDS-bandwidth bound (one ds_read_b32 per
// multiply-add: 160 us for the 10 240 17x17 decoder tiles); an MFMA consumes two LDS dwords per lane for 16 FMAs.
// One workgroup per pair, one wave per 16-row query tile.  LDS pitches: operands read as [row = M/N index][k]
// use a pitch = 4 (mod 32) dwords, operands read as [k][n] use a pitch = 16 (mod 32): both conflict-free.
constexpr int PK_ = 66, PN_ = 80;   // 66 = 2 (mod 32): 16 rows x 2 k-lanes of a half-wave hit 32 distinct banks
static int attn_impl() { return ortk::tuning().attn_impl; }
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LKP_T / DKP_T (/ LQP_T): padded sizes as compile-time constants for the path's shapes (0 = take the run-time value):
// with constant trip counts the MFMA loops unroll fully and their LDS reads are issued ahead of the MFMA chain.
template <int LKP_T, int DKP_T>
__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(ortk_attn_args a, int Lkp_rt, int DKP_rt) {
    const int Lkp = LKP_T > 0 ? LKP_T : Lkp_rt, DKP = DKP_T > 0 ? DKP_T : DKP_rt;
    extern __shared__ __attribute__((aligned(16))) float smem_m[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int Lk = a.Lk, dk = a.dk, Lq = a.Lq;
    float* sK = smem_m;                         // [Lkp][PK_]  rows = key,  k = feature
    float* sV = sK + Lkp * PK_;                 // [Lkp][PN_]  rows = key (k of P.V), n = feature
    float* sMask = sV + Lkp * PN_;              // [64]
    float* sQ = sMask + 64 + wave * (16 * PK_ * 2);   // [16][PK_]
    float* sP = sQ + 16 * PK_;                  // [16][PK_]   rows = query, k = key
    // dk % 4 == 0 and 16-byte aligned rows (checked by the launcher): stage with float4 loads
    for (int idx = tid; idx < Lkp * (DKP / 4); idx += blockDim.x) {
        const int j = idx / (DKP / 4), dd = (idx - j * (DKP / 4)) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (j < Lk && dd < dk) {
            const int64_t row = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * Lk + j]
                                           : (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : Lk) + j;
            kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * dk + dd);
            vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * dk + dd);
        }
        float* pk = sK + j * PK_ + dd; pk[0] = kv.x; pk[1] = kv.y; pk[2] = kv.z; pk[3] = kv.w;
        *reinterpret_cast<float4*>(sV + j * PN_ + dd) = vv;
    }
    if (tid < 64) sMask[tid] = (tid < Lk) ? (a.kmask ? a.kmask[(int64_t)g * Lk + tid] : 1.f) : -1.f;   // -1: padded key
    __syncthreads();
    const float scale = sqrtf((float)dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int lr = lane & 15, lq = lane >> 4;
    const int njt = Lkp >> 4, ndt = DKP >> 4, nit = (Lq + 15) >> 4;
    for (int it = wave; it < nit; it += nw) {
        const int i0 = it * 16;
        for (int idx = lane; idx < 16 * (DKP / 4); idx += 64) {
            const int r = idx / (DKP / 4), dd = (idx - r * (DKP / 4)) * 4;
            const int i = i0 + r;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < Lq && dd < dk) qv = *reinterpret_cast<const float4*>(a.q + ((int64_t)g * Lq + i) * a.ldq + h * dk + dd);
            float* pq = sQ + r * PK_ + dd; pq[0] = qv.x; pq[1] = qv.y; pq[2] = qv.z; pq[3] = qv.w;
        }
        wave_sync();
        // S = Q K^T : D[i = 4*lq + r][j = 16*jt + lr]
        f4 sacc[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            sacc[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            if (jt < njt) {
#pragma unroll
                for (int kk = 0; kk < DKP / 4; ++kk) {
                    const float av = sQ[lr * PK_ + 4 * kk + lq];
                    const float bv = sK[(16 * jt + lr) * PK_ + 4 * kk + lq];
                    sacc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, sacc[jt], 0, 0, 0);
                }
            }
        }
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < njt) {
                const int j = 16 * jt + lr;
                const float mk = sMask[j];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    float x = sacc[jt][r] / scale;
                    const int qpos = a.causal_period > 0 ? i % a.causal_period : 0;
                    if (mk == 0.f || (a.causal_period > 0 && j > qpos)) x = -1e9f;
                    if (mk < 0.f) x = -INFINITY;                                   // padded key: not part of the row
                    else if (a.bias && i < Lq) x = a.bias[(((int64_t)g * a.H + h) * Lq + i) * Lk + j] + x;
                    sacc[jt][r] = x;
                    mx[r] = fmaxf(mx[r], x);
                }
            }
        }
        float sum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { mx[r] = group16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt < njt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = expf(sacc[jt][r] - mx[r]); sacc[jt][r] = e; sum[r] += e; }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = group16_sum(sum[r]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < njt) {
                const int j = 16 * jt + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    float p = sacc[jt][r] / sum[r];
                    if (i < Lq && j < Lk) {
                        const int64_t pi = (((int64_t)g * a.H + h) * Lq + i) * Lk + j;
                        if (a.p) a.p[pi] = p;
                        if (a.drop_p > 0.f) p = ortk_keep(a.drop_seed, (uint64_t)pi, a.drop_p) ? p * inv_keep : 0.f;
                    } else p = 0.f;
                    sP[(4 * lq + r) * PK_ + j] = p;
                }
            }
        }
        wave_sync();
        // O = P V : D[i = 4*lq + r][d = 16*dt + lr]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (dt < ndt) {
                f4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < Lkp / 4; ++kk) {
                    const float av = sP[lr * PK_ + 4 * kk + lq];
                    const float bv = sV[(4 * kk + lq) * PN_ + 16 * dt + lr];
                    oacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, oacc, 0, 0, 0);
                }
                const int dd = 16 * dt + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    if (i < Lq && dd < dk) st_elem(a.o, ((int64_t)g * Lq + i) * a.ldo + h * dk + dd, a.o_dtype, oacc[r]);
                }
            }
        }
        wave_sync();
    }
}

// Backward with the same tiling.  Phase 1 (wave = 16-row query tile): dP = dO V^T, dS = P (dP - rowsum(P dP)),
// dQ = dS K / sqrt(dk); dS / sqrt(dk) and the dropped P go to workgroup-wide LDS arrays.  Phase 2 (waves share the
// 2 x (Lkp/16) x (DKP/16) output tiles): dK = dS^T Q / sqrt(dk), dV = Pd^T dO over ALL query rows of the group.
// PART: 0 = everything; 1 = dQ (and dscore) only — phase 1 alone, no Q / dropped-P images: 79 KB of LDS at the cross-
// attention shape, two workgroups per CU; 2 = dK and dV only (phase 1 without the dQ product, then phase 2).  The
// executor runs part 1 on the critical path and part 2, whose results are only needed for the cross-attention K/V
// projection at the end of the decoder backward, on its side stream.
template <int LKP_T, int DKP_T, int LQP_T, int PART>
__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(ortk_attn_args a, int Lkp_rt, int DKP_rt, int Lqp_rt) {
    const int Lkp = LKP_T > 0 ? LKP_T : Lkp_rt, DKP = DKP_T > 0 ? DKP_T : DKP_rt, Lqp = LQP_T > 0 ? LQP_T : Lqp_rt;
    extern __shared__ __attribute__((aligned(16))) float smem_m[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int Lk = a.Lk, dk = a.dk, Lq = a.Lq;
    float* sK = smem_m;                 // [Lkp][PN_]  read as [k = key][n = feature]      (dQ = dS K)
    float* sV = sK + Lkp * PN_;         // [Lkp][PK_]  read as [n = key][k = feature]      (dP = dO V^T)
    float* sQ = sV + Lkp * PK_;         // [Lqp][PN_]  read as [k = query][n = feature]    (dK = dS^T Q)       (not PART 1)
    float* sG = PART == 1 ? sQ : sQ + Lqp * PN_;   // [Lqp][PK_]  dO: A operand of dP, [k][n] operand of dV
    float* sS = sG + Lqp * PK_;         // [Lqp][PK_]  dS / sqrt(dk)
    float* sD = sS + Lqp * PK_;         // [Lqp][PK_]  dropped P                                           (not PART 1)
    for (int idx = tid; idx < Lkp * (DKP / 4); idx += blockDim.x) {
        const int j = idx / (DKP / 4), dd = (idx - j * (DKP / 4)) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (j < Lk && dd < dk) {
            const int64_t row = (int64_t)g * Lk + j;
            kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * dk + dd);
            vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * dk + dd);
        }
        *reinterpret_cast<float4*>(sK + j * PN_ + dd) = kv;
        float* pv = sV + j * PK_ + dd; pv[0] = vv.x; pv[1] = vv.y; pv[2] = vv.z; pv[3] = vv.w;
    }
    for (int idx = tid; idx < Lqp * (DKP / 4); idx += blockDim.x) {
        const int i = idx / (DKP / 4), dd = (idx - i * (DKP / 4)) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
        if (i < Lq && dd < dk) {
            const int64_t qrow = (int64_t)g * Lq + i;
            qv = *reinterpret_cast<const float4*>(a.q + qrow * a.ldq + h * dk + dd);
            gv = *reinterpret_cast<const float4*>(a.d_o + qrow * a.lddo + h * dk + dd);
        }
        if (PART != 1) *reinterpret_cast<float4*>(sQ + i * PN_ + dd) = qv;
        float* pg = sG + i * PK_ + dd; pg[0] = gv.x; pg[1] = gv.y; pg[2] = gv.z; pg[3] = gv.w;
    }
    __syncthreads();
    const float scale = sqrtf((float)dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int lr = lane & 15, lq = lane >> 4;
    const int njt = Lkp >> 4, ndt = DKP >> 4, nit = Lqp >> 4;
    for (int it = wave; it < nit; it += nw) {
        const int i0 = it * 16;
        // dP = dO V^T : D[i = 4*lq + r][j = 16*jt + lr]
        f4 dp[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            dp[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            if (jt < njt)
#pragma unroll
                for (int kk = 0; kk < DKP / 4; ++kk) {
                    const float av = sG[(i0 + lr) * PK_ + 4 * kk + lq];
                    const float bv = sV[(16 * jt + lr) * PK_ + 4 * kk + lq];
                    dp[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, dp[jt], 0, 0, 0);
                }
        }
        f4 pp[4];
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            pp[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            if (jt < njt) {
                const int j = 16 * jt + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    float p = 0.f, d = 0.f, pd = 0.f;
                    if (i < Lq && j < Lk) {
                        const int64_t pi = (((int64_t)g * a.H + h) * Lq + i) * Lk + j;
                        p = a.p[pi];
                        const bool keep = a.drop_p > 0.f ? ortk_keep(a.drop_seed, (uint64_t)pi, a.drop_p) : true;
                        d = keep ? dp[jt][r] * inv_keep : 0.f;
                        pd = keep ? p * inv_keep : 0.f;
                    }
                    pp[jt][r] = p; dp[jt][r] = d; dot[r] += p * d;
                    if (PART != 1) sD[(i0 + 4 * lq + r) * PK_ + j] = pd;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dot[r] = group16_sum(dot[r]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < njt) {
                const int j = 16 * jt + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    const float ds = pp[jt][r] * (dp[jt][r] - dot[r]);
                    if (PART != 2 && a.dscore && i < Lq && j < Lk) a.dscore[(((int64_t)g * a.H + h) * Lq + i) * Lk + j] = ds;
                    sS[(i0 + 4 * lq + r) * PK_ + j] = ds / scale;
                }
            }
        }
        wave_sync();
        // dQ = dS K : D[i][d]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (PART != 2 && dt < ndt) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < Lkp / 4; ++kk) {
                    const float av = sS[(i0 + lr) * PK_ + 4 * kk + lq];
                    const float bv = sK[(4 * kk + lq) * PN_ + 16 * dt + lr];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
                const int dd = 16 * dt + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lq + r;
                    if (i < Lq && dd < dk) st_elem(a.dq, ((int64_t)g * Lq + i) * a.lddq + h * dk + dd, a.dqkv_dtype, acc[r]);
                }
            }
        }
    }
    if (PART == 1) return;
    __syncthreads();
    // phase 2: output tiles t = (which, jt, dt); A[m = key][k = query] = S^T (read [k][n]-style from sS / sD rows = query)
    const int ntiles = 2 * njt * ndt;
    for (int t = wave; t < ntiles; t += nw) {
        const int which = t / (njt * ndt), rem = t - which * njt * ndt, jt = rem / ndt, dt = rem - jt * ndt;
        const float* sA = which == 0 ? sS : sD;      // [query][key]
        const float* sB = which == 0 ? sQ : sG;      // [query][feature]
        const int pb = which == 0 ? PN_ : PK_;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < Lqp / 4; ++kk) {
            const float av = sA[(4 * kk + lq) * PK_ + 16 * jt + lr];    // A[m = 16*jt + lr][k = 4*kk + lq]
            const float bv = sB[(4 * kk + lq) * pb + 16 * dt + lr];     // B[k][n = 16*dt + lr]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        const int dd = 16 * dt + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * lq + r;
            if (j < Lk && dd < dk) {
                const int64_t row = (int64_t)g * Lk + j;
                if (which == 0) st_elem(a.d_k, row * a.lddk + h * dk + dd, a.dqkv_dtype, acc[r]);
                else            st_elem(a.dv, row * a.lddv + h * dk + dd, a.dqkv_dtype, acc[r]);
            }
        }
    }
}

// K / V rows may be stored as bf16 (decode-time caches in mixed precision: ortk_attn_args.kv_dtype = 1)
template <typename T> __device__ __forceinline__ void ld_row16(const T* p, float4 (&o)[4]);
template <> __device__ __forceinline__ void ld_row16<float>(const float* p, float4 (&o)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = *reinterpret_cast<const float4*>(p + 4 * c);
}
template <> __device__ __forceinline__ void ld_row16<__bf16>(const __bf16* p, float4 (&o)[4]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p), b = *reinterpret_cast<const bf16x8*>(p + 8);
    o[0] = make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]); o[1] = make_float4((float)a[4], (float)a[5], (float)a[6], (float)a[7]);
    o[2] = make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]); o[3] = make_float4((float)b[4], (float)b[5], (float)b[6], (float)b[7]);
}
template <typename T> __device__ __forceinline__ void ld_row8(const T* p, float4& lo, float4& hi);
template <> __device__ __forceinline__ void ld_row8<float>(const float* p, float4& lo, float4& hi) {
    lo = *reinterpret_cast<const float4*>(p); hi = *reinterpret_cast<const float4*>(p + 4);
}
template <> __device__ __forceinline__ void ld_row8<__bf16>(const __bf16* p, float4& lo, float4& hi) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
    lo = make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]); hi = make_float4((float)a[4], (float)a[5], (float)a[6], (float)a[7]);
}

// ------------------------------------------------------------------------------------------------
// Register-only form for short sequences (Lq, Lk <= 32, dk = 64): the decoder's 17 x 17 causal self-attention is
// 10 240 tiny (caption, head) pairs per layer.  One wave per pair, no LDS, no barrier: every MFMA operand is loaded
// from global memory straight into the lane that feeds it.
//   * the feature sum of Q.K^T is order-free, so lane (lr, lq) feeds features 16*lq .. 16*lq+15 of row lr over the
//     16 k-steps: four float4 loads per row instead of a transposing LDS image;
//   * the scores are produced TRANSPOSED (S^T = K Q^T): lane (lr = query i, lq) then holds keys j = 16*jt + 4*lq + r,
//     which is exactly the A-operand layout of P.V (k-step (jt, r) takes key 16*jt + 4*lq + r from every lane), so
//     the probabilities never leave their registers; the matching V rows are read as 64-byte row segments.
// The backward runs the same scheme twice: transposed layout for dQ, natural layout (operands of the dP product
// swapped, no extra loads) for dK and dV.
template <int NIT, int NJT, typename KVT>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(ortk_attn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nkv * a.H) return;
    const int g = pair / a.H, h = pair - g * a.H;
    const int Lq = a.Lq, Lk = a.Lk;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t kv0 = (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : Lk);
    const float* qb = a.q + (int64_t)g * Lq * a.ldq + h * 64;
    const KVT* kb = reinterpret_cast<const KVT*>(a.k) + kv0 * a.ldk + h * 64;
    const KVT* vb = reinterpret_cast<const KVT*>(a.v) + kv0 * a.ldv + h * 64;
    float4 kf[NJT][4], qf[NIT][4];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) ld_row16<KVT>(kb + (int64_t)min(16 * jt + lr, Lk - 1) * a.ldk + 16 * lq, kf[jt]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float* src = qb + (int64_t)min(16 * it + lr, Lq - 1) * a.ldq + 16 * lq;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[it][c] = *reinterpret_cast<const float4*>(src + 4 * c);
    }
    // key mask of this lane's keys j = 16*jt + 4*lq + r
    float mk[NJT][4];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * lq + r;
            mk[jt][r] = j < Lk ? (a.kmask ? a.kmask[(int64_t)g * Lk + j] : 1.f) : -1.f;
        }
    const float scale = sqrtf(64.f);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    f4 pt[NIT][NJT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        // S^T[j][i]: A = K rows (m = j), B = Q rows (n = i)
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][c].x, qf[it][c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][c].y, qf[it][c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][c].z, qf[it][c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][c].w, qf[it][c].w, acc, 0, 0, 0);
            }
            pt[it][jt] = acc;
        }
        const int i = 16 * it + lr;
        const int qpos = a.causal_period > 0 ? i % a.causal_period : 0;
        const int64_t prow = (((int64_t)g * a.H + h) * Lq + min(i, Lq - 1)) * Lk;
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lq + r;
                float x = pt[it][jt][r] / scale;
                if (mk[jt][r] == 0.f || (a.causal_period > 0 && j > qpos)) x = -1e9f;
                if (mk[jt][r] < 0.f) x = -INFINITY;
                else if (a.bias) x = a.bias[prow + j] + x;
                pt[it][jt][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = expf(pt[it][jt][r] - mx); pt[it][jt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lq + r;
                float pv = pt[it][jt][r] / sum;
                if (i < Lq && j < Lk) {
                    if (a.p) a.p[prow + j] = pv;
                    if (a.drop_p > 0.f) pv = ortk_keep(a.drop_seed, (uint64_t)(prow + j), a.drop_p) ? pv * inv_keep : 0.f;
                } else pv = 0.f;
                pt[it][jt][r] = pv;
            }
    }
    // O[i][d] = sum_j P[i][j] V[j][d]: A = P (held), B = V[16*jt + 4*lq + r][16*dt + lr]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float vv[NJT][4];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lq + r;
                const float t = (float)vb[(int64_t)min(j, Lk - 1) * a.ldv + 16 * dt + lr];
                vv[jt][r] = j < Lk ? t : 0.f;
            }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pt[it][jt][r], vv[jt][r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * lq + r;
                if (i < Lq) st_elem(a.o, ((int64_t)g * Lq + i) * a.ldo + h * 64 + 16 * dt + lr, a.o_dtype, acc[r]);
            }
        }
    }
}

template <int NIT, int NJT>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(ortk_attn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nkv * a.H) return;
    const int g = pair / a.H, h = pair - g * a.H;
    const int Lq = a.Lq, Lk = a.Lk;
    const int lr = lane & 15, lq = lane >> 4;
    const float* qb = a.q + (int64_t)g * Lq * a.ldq + h * 64;
    const float* kb = a.k + (int64_t)g * Lk * a.ldk + h * 64;
    const float* vb = a.v + (int64_t)g * Lk * a.ldv + h * 64;
    const float* gb = a.d_o + (int64_t)g * Lq * a.lddo + h * 64;
    const int64_t pbase = ((int64_t)g * a.H + h) * Lq * Lk;
    float4 vf[NJT][4], gf[NIT][4];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const float* src = vb + (int64_t)min(16 * jt + lr, Lk - 1) * a.ldv + 16 * lq;
#pragma unroll
        for (int c = 0; c < 4; ++c) vf[jt][c] = *reinterpret_cast<const float4*>(src + 4 * c);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float* src = gb + (int64_t)min(16 * it + lr, Lq - 1) * a.lddo + 16 * lq;
#pragma unroll
        for (int c = 0; c < 4; ++c) gf[it][c] = *reinterpret_cast<const float4*>(src + 4 * c);
    }
    const float scale = sqrtf(64.f);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    float dotv[NIT];
    // ---- transposed layout: lane (lr = i, lq) holds keys j = 16*jt + 4*lq + r   ->  dQ = dS K / sqrt(dk)
    {
        f4 ds[NIT][NJT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = 16 * it + lr;
            f4 pp[NJT];
            float dot = 0.f;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};   // dP^T[j][i] = sum_d V[j][d] dO[i][d]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[jt][c].x, gf[it][c].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[jt][c].y, gf[it][c].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[jt][c].z, gf[it][c].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[jt][c].w, gf[it][c].w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * lq + r;
                    float pv = 0.f, d = 0.f;
                    if (i < Lq && j < Lk) {
                        const int64_t pi = pbase + (int64_t)i * Lk + j;
                        pv = a.p[pi];
                        const bool keep = a.drop_p > 0.f ? ortk_keep(a.drop_seed, (uint64_t)pi, a.drop_p) : true;
                        d = keep ? acc[r] * inv_keep : 0.f;
                    }
                    dot += pv * d;
                    pp[jt][r] = pv; ds[it][jt][r] = d;
                }
            }
            dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
            dotv[it] = dot;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * lq + r;
                    const float v = pp[jt][r] * (ds[it][jt][r] - dot);
                    if (a.dscore && i < Lq && j < Lk) a.dscore[pbase + (int64_t)i * Lk + j] = v;
                    ds[it][jt][r] = v / scale;
                }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float kk[NJT][4];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) kk[jt][r] = kb[(int64_t)min(16 * jt + 4 * lq + r, Lk - 1) * a.ldk + 16 * dt + lr];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[it][jt][r], kk[jt][r], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + 4 * lq + r;
                    if (i < Lq) st_elem(a.dq, ((int64_t)g * Lq + i) * a.lddq + h * 64 + 16 * dt + lr, a.dqkv_dtype, acc[r]);
                }
            }
        }
    }
    // ---- natural layout: lane (lr = j, lq) holds queries i = 16*it + 4*lq + r   ->  dK = dS^T Q / sqrt(dk), dV = Pd^T dO
    f4 ds[NJT][NIT], pd[NJT][NIT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int j = 16 * jt + lr;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};   // dP[i][j] = sum_d dO[i][d] V[j][d]  (same operand registers, swapped)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[it][c].x, vf[jt][c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[it][c].y, vf[jt][c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[it][c].z, vf[jt][c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[it][c].w, vf[jt][c].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * lq + r;
                const float dot = __shfl(dotv[it], 4 * lq + r, 64);     // lane (lr' = i mod 16, lq' = 0) holds row i's dot
                float pv = 0.f, d = 0.f, pdv = 0.f;
                if (i < Lq && j < Lk) {
                    const int64_t pi = pbase + (int64_t)i * Lk + j;
                    pv = a.p[pi];
                    const bool keep = a.drop_p > 0.f ? ortk_keep(a.drop_seed, (uint64_t)pi, a.drop_p) : true;
                    d = keep ? acc[r] * inv_keep : 0.f;
                    pdv = keep ? pv * inv_keep : 0.f;
                }
                ds[jt][it][r] = pv * (d - dot) / scale;
                pd[jt][it][r] = pdv;
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float qq[NIT][4], gg[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = min(16 * it + 4 * lq + r, Lq - 1);
                qq[it][r] = qb[row * a.ldq + 16 * dt + lr];
                gg[it][r] = gb[row * a.lddo + 16 * dt + lr];
            }
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            f4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ak = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[jt][it][r], qq[it][r], ak, 0, 0, 0);
                    av = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[jt][it][r], gg[it][r], av, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lq + r;
                if (j < Lk) {
                    const int64_t row = (int64_t)g * Lk + j;
                    st_elem(a.d_k, row * a.lddk + h * 64 + 16 * dt + lr, a.dqkv_dtype, ak[r]);
                    st_elem(a.dv, row * a.lddv + h * 64 + 16 * dt + lr, a.dqkv_dtype, av[r]);
                }
            }
        }
    }
}

// Decode-time self-attention (ONE query row per K/V group, H = 8, dk = 64): one wave per row serves all 8 heads.
// Lane l owns features 8l .. 8l+7 of the 512-wide row (head l / 8), so every K / V row of the cache is ONE coalesced
// 2-KB read shared by the 8 heads; a score is 8 FMAs + a 3-step reduction over the 8 lanes of the head; the soft-max
// runs redundantly in those lanes; P.V is 8 FMAs per key.  Keys come through the beam ancestry table (kv_index) or
// a fixed stride.  Replaces 8 x rows workgroups of the generic block kernel (75 us -> see DESIGN.md section 7).
template <int LKMAX, typename KVT>
__global__ __launch_bounds__(256) void attn_rowdec_kernel(ortk_attn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wave;
    if (g >= a.nkv) return;
    const int Lk = a.Lk;
    const float* qp = a.q + (int64_t)g * a.ldq + 8 * lane;
    const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
    const int64_t base = (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : Lk);
    float sc[LKMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LKMAX; ++j) {
        sc[j] = -INFINITY;
        if (j < Lk) {
            const int64_t row = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * Lk + j] : base + j;
            float4 k0, k1;
            if (a.k_new && j == Lk - 1) {      // the new position: take K from the projection output and append it to the cache
                ld_row8<float>(a.k_new + (int64_t)g * a.ld_new + 8 * lane, k0, k1);
                st_elem4(const_cast<float*>(a.k), row * a.ldk + 8 * lane, Elem<KVT>::DT, k0);
                st_elem4(const_cast<float*>(a.k), row * a.ldk + 8 * lane + 4, Elem<KVT>::DT, k1);
            } else {
                ld_row8<KVT>(reinterpret_cast<const KVT*>(a.k) + row * a.ldk + 8 * lane, k0, k1);
            }
            float d = q0.x * k0.x + q0.y * k0.y + q0.z * k0.z + q0.w * k0.w + q1.x * k1.x + q1.y * k1.y + q1.z * k1.z + q1.w * k1.w;
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            d *= 0.125f;                                              // 1 / sqrt(64)
            if (a.kmask && a.kmask[(int64_t)g * Lk + j] == 0.f) d = -1e9f;
            sc[j] = d;
            mx = fmaxf(mx, d);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LKMAX; ++j) { const float e = j < Lk ? expf(sc[j] - mx) : 0.f; sc[j] = e; sum += e; }
    const float inv = 1.f / sum;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LKMAX; ++j) {
        if (j < Lk) {
            const int64_t row = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * Lk + j] : base + j;
            float4 v0, v1;
            if (a.v_new && j == Lk - 1) {
                ld_row8<float>(a.v_new + (int64_t)g * a.ld_new + 8 * lane, v0, v1);
                st_elem4(const_cast<float*>(a.v), row * a.ldv + 8 * lane, Elem<KVT>::DT, v0);
                st_elem4(const_cast<float*>(a.v), row * a.ldv + 8 * lane + 4, Elem<KVT>::DT, v1);
            } else {
                ld_row8<KVT>(reinterpret_cast<const KVT*>(a.v) + row * a.ldv + 8 * lane, v0, v1);
            }
            const float pj = sc[j] * inv;
            if (a.p) { if ((lane & 7) == 0) a.p[((int64_t)g * a.H + (lane >> 3)) * Lk + j] = pj; }
            o[0] += pj * v0.x; o[1] += pj * v0.y; o[2] += pj * v0.z; o[3] += pj * v0.w;
            o[4] += pj * v1.x; o[5] += pj * v1.y; o[6] += pj * v1.z; o[7] += pj * v1.w;
        }
    }
    const int64_t oi = (int64_t)g * a.ldo + 8 * lane;
    st_elem4(a.o, oi, a.o_dtype, make_float4(o[0], o[1], o[2], o[3]));
    st_elem4(a.o, oi + 4, a.o_dtype, make_float4(o[4], o[5], o[6], o[7]));
}

// Decode form (few query rows per K/V group: 1 new token per beam for self-attention, the beams of an image for
// cross-attention).  One wave per (group, head); K and V of the pair live in REGISTERS — K with lane = key (each
// lane holds its key's 64 features, loaded as 16 independent float4), V with lane = feature (one coalesced 256-B row
// per key) — so nothing is staged in LDS and every load of the pair is in flight at once.  Per query: scores are 64
// FMAs per lane against the broadcast query, soft-max statistics are two wave reductions, P.V is Lk shuffle+FMA.
// (A first version reduced one key at a time behind per-key branches: 250 us per call, all exposed latency.)
template <int LKMAX>
__global__ __launch_bounds__(256) void attn_decode_kernel(ortk_attn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nkv * a.H) return;
    const int g = pair / a.H, h = pair - g * a.H;
    const int Lk = a.Lk, dk = a.dk, Lq = a.Lq;
    const int jl = lane < Lk ? lane : Lk - 1;
    const int64_t myrow = a.kv_index ? (int64_t)a.kv_index[(int64_t)g * Lk + jl]
                                     : (int64_t)g * (a.kv_group_stride > 0 ? a.kv_group_stride : Lk) + jl;
    // K: lane = key
    float4 kreg[MAXD / 4];
    const float* kp = a.k + myrow * a.ldk + h * dk;
#pragma unroll
    for (int c = 0; c < MAXD / 4; ++c) kreg[c] = (4 * c < dk) ? *reinterpret_cast<const float4*>(kp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    // V: lane = feature; the row of key j is fetched through lane j's row index
    float vreg[LKMAX];
    const int dl = lane < dk ? lane : 0;
#pragma unroll
    for (int j = 0; j < LKMAX; ++j) {
        const int64_t row = __shfl(myrow, j < Lk ? j : 0, 64);
        vreg[j] = (j < Lk) ? a.v[row * a.ldv + h * dk + dl] : 0.f;
    }
    const float kmask = (lane < Lk && a.kmask) ? a.kmask[(int64_t)g * Lk + lane] : 1.f;
    const float scale = sqrtf((float)dk);
    for (int i = 0; i < Lq; ++i) {
        const int64_t qrow = (int64_t)g * Lq + i;
        const float* qp = a.q + qrow * a.ldq + h * dk;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < MAXD / 4; ++c) {
            if (4 * c < dk) {
                const float4 q4 = *reinterpret_cast<const float4*>(qp + 4 * c);      // wave-uniform address: broadcast
                acc += q4.x * kreg[c].x + q4.y * kreg[c].y + q4.z * kreg[c].z + q4.w * kreg[c].w;
            }
        }
        float sc = -INFINITY;
        const int64_t pbase = (((int64_t)g * a.H + h) * Lq + i) * Lk;
        if (lane < Lk) {
            acc = acc / scale;
            if (kmask == 0.f) acc = -1e9f;
            if (a.bias) acc = a.bias[pbase + lane] + acc;
            sc = acc;
        }
        const float mx = wave_max(sc);
        const float e = lane < Lk ? expf(sc - mx) : 0.f;
        const float p = e / wave_sum(e);
        if (a.p && lane < Lk) a.p[pbase + lane] = p;
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < LKMAX; ++j) o += __shfl(p, j, 64) * vreg[j];      // p is 0 beyond Lk
        if (lane < dk) st_elem(a.o, qrow * a.ldo + h * dk + lane, a.o_dtype, o);
    }
}

struct TilesB {
    float (*k)[KP]; float (*v)[KP]; float (*dk)[KP]; float (*dv)[KP];   // [Lk][KP] each
    float (*q)[MAXD]; float (*go)[MAXD];                                // [4][MAXD]
    float (*ds)[MAXK]; float (*pd)[MAXK];                               // [4][MAXK]
};
__host__ __device__ inline size_t bwd_lds_bytes(int Lk) {
    return sizeof(float) * ((size_t)4 * Lk * KP + 8 * MAXD + 8 * MAXK);
}
__device__ __forceinline__ TilesB carve_bwd(float* base, int Lk) {
    TilesB t;
    t.k = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.v = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.dk = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.dv = reinterpret_cast<float (*)[KP]>(base); base += (size_t)Lk * KP;
    t.q = reinterpret_cast<float (*)[MAXD]>(base); base += 4 * MAXD;
    t.go = reinterpret_cast<float (*)[MAXD]>(base); base += 4 * MAXD;
    t.ds = reinterpret_cast<float (*)[MAXK]>(base); base += 4 * MAXK;
    t.pd = reinterpret_cast<float (*)[MAXK]>(base);
    return t;
}

// Backward: dP = dO V^T (through the dropout mask), dS = P (dP - rowsum(P dP)), dQ = dS K / sqrt(dk),
// dK = dS^T Q / sqrt(dk), dV = Pdrop^T dO.  dscore (optional) receives dS — the gradient of the additive
// geometry bias (and of the pre-softmax scores).
//
// Fast form (Lk <= 64): ONE WAVE owns one (K/V group, head) pair — no workgroup barrier, no cross-wave reduction.
// lane = key for the score-side math, lane = feature for the dQ / dK / dV side.  The wave keeps dK[j][lane] and
// dV[j][lane] for ALL keys of its group in registers while it walks the group's query rows and writes them once
// at the end (256-B coalesced rows).  Per-row scalars cross lanes through a 3 x 64-float LDS scratch line
// (broadcast reads).  K and V of the pair sit in the wave's own LDS slice (pitch dk+1).
// History: v1 accumulated dK/dV with ds_add_f32 atomics (53 % of the training step); v2 used one workgroup per pair
// with a 4-phase LDS reduction (10 ms/step, mostly barriers and idle waves on 17-row tiles).
template <int LKMAX>
__global__ __launch_bounds__(256) void attn_bwd_wave_kernel(ortk_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_b[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nkv * a.H) return;                 // whole wave exits; no barriers below
    const int g = pair / a.H, h = pair - g * a.H;
    const int Lk = a.Lk, dk = a.dk;
    const size_t per_wave = (size_t)2 * Lk * KP + 3 * 64;
    float* base = smem_b + wave * per_wave;
    float (*sK)[KP] = reinterpret_cast<float (*)[KP]>(base);
    float (*sV)[KP] = reinterpret_cast<float (*)[KP]>(base + (size_t)Lk * KP);
    float* sgo = base + (size_t)2 * Lk * KP;         // dO row       (lane = feature)
    float* sds = sgo + 64;                           // dS / sqrt(dk) (lane = key)
    float* spd = sds + 64;                           // dropped P     (lane = key)
    for (int idx = lane; idx < Lk * KP; idx += 64) {
        const int j = idx / KP, dd = idx - j * KP;
        const int64_t row = (int64_t)g * Lk + j;
        const bool in = dd < dk;
        sK[j][dd] = in ? a.k[row * a.ldk + h * dk + dd] : 0.f;
        sV[j][dd] = in ? a.v[row * a.ldv + h * dk + dd] : 0.f;
    }
    wave_sync();
    const float scale = sqrtf((float)dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    float acck[LKMAX], accv[LKMAX];
#pragma unroll
    for (int j = 0; j < LKMAX; ++j) { acck[j] = 0.f; accv[j] = 0.f; }
    const int jl = lane < Lk ? lane : 0, dl = lane < dk ? lane : 0;
    for (int i = 0; i < a.Lq; ++i) {
        const int64_t qrow = (int64_t)g * a.Lq + i;
        float qv = 0.f, gv = 0.f;                      // lane = feature
        if (lane < dk) {
            qv = a.q[qrow * a.ldq + h * dk + lane];
            gv = a.d_o[qrow * a.lddo + h * dk + lane];
        }
        sgo[lane] = gv;
        const int64_t pbase = (((int64_t)g * a.H + h) * a.Lq + i) * Lk;
        float p = 0.f;
        bool keep = true;
        if (lane < Lk) {
            p = a.p[pbase + lane];
            if (a.drop_p > 0.f) keep = ortk_keep(a.drop_seed, (uint64_t)(pbase + lane), a.drop_p);
        }
        wave_sync();
        // lane = key: dP_j = sum_d dO_d V[j][d]   (pad features hold zeros)
        float acc = 0.f;
#pragma unroll 8
        for (int dd = 0; dd < MAXD; ++dd) acc += sgo[dd] * sV[jl][dd];
        const float dp = (lane < Lk && keep) ? acc * inv_keep : 0.f;
        const float pd = (lane < Lk && keep) ? p * inv_keep : 0.f;
        const float dot = wave_sum(p * dp);
        const float ds = p * (dp - dot);
        if (a.dscore && lane < Lk) a.dscore[pbase + lane] = ds;
        sds[lane] = ds / scale;
        spd[lane] = pd;
        wave_sync();
        // lane = feature: dQ_d = sum_j dS_j K[j][d];  dK[j][d] += dS_j q_d;  dV[j][d] += Pd_j dO_d
        float dq = 0.f;
#pragma unroll
        for (int j = 0; j < LKMAX; ++j) {
            if (j < Lk) {
                const float dsj = sds[j], pdj = spd[j];
                dq += dsj * sK[j][dl];
                acck[j] += dsj * qv;
                accv[j] += pdj * gv;
            }
        }
        if (lane < dk) st_elem(a.dq, qrow * a.lddq + h * dk + lane, a.dqkv_dtype, dq);
        wave_sync();
    }
    if (lane < dk) {
#pragma unroll
        for (int j = 0; j < LKMAX; ++j) {
            if (j < Lk) {
                const int64_t row = (int64_t)g * Lk + j;
                st_elem(a.d_k, row * a.lddk + h * dk + lane, a.dqkv_dtype, acck[j]);
                st_elem(a.dv, row * a.lddv + h * dk + lane, a.dqkv_dtype, accv[j]);
            }
        }
    }
}

// General form (64 < Lk <= 128, e.g. images with up to 100 regions): LDS accumulation with ds_add_f32.
__global__ __launch_bounds__(256) void attn_bwd_kernel(ortk_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_b[];
    TilesB t = carve_bwd(smem_b, a.Lk);
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    for (int idx = tid; idx < a.Lk * a.dk; idx += 256) {
        const int j = idx / a.dk, dd = idx - j * a.dk;
        const int64_t row = (int64_t)g * a.Lk + j;
        t.k[j][dd] = a.k[row * a.ldk + h * a.dk + dd];
        t.v[j][dd] = a.v[row * a.ldv + h * a.dk + dd];
        t.dk[j][dd] = 0.f;
        t.dv[j][dd] = 0.f;
    }
    __syncthreads();
    const float scale = sqrtf((float)a.dk);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    for (int i = wave; i < a.Lq; i += 4) {
        const int64_t qrow = (int64_t)g * a.Lq + i;
        if (lane < a.dk) {
            t.q[wave][lane] = a.q[qrow * a.ldq + h * a.dk + lane];
            t.go[wave][lane] = a.d_o[qrow * a.lddo + h * a.dk + lane];
        }
        wave_sync();
        const int64_t pbase = (((int64_t)g * a.H + h) * a.Lq + i) * a.Lk;
        float p[2], dp[2], dot = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            p[u] = 0.f; dp[u] = 0.f;
            if (j < a.Lk) {
                p[u] = a.p[pbase + j];
                float acc = 0.f;
                for (int dd = 0; dd < a.dk; ++dd) acc += t.go[wave][dd] * t.v[j][dd];
                const bool keep = a.drop_p > 0.f ? ortk_keep(a.drop_seed, (uint64_t)(pbase + j), a.drop_p) : true;
                dp[u] = keep ? acc * inv_keep : 0.f;
                t.pd[wave][j] = keep ? p[u] * inv_keep : 0.f;
                dot += p[u] * dp[u];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < a.Lk) {
                const float ds = p[u] * (dp[u] - dot);
                if (a.dscore) a.dscore[pbase + j] = ds;
                t.ds[wave][j] = ds / scale;
            }
        }
        wave_sync();
        if (lane < a.dk) {
            float dq = 0.f;
            const float qv = t.q[wave][lane], gv = t.go[wave][lane];
            for (int j = 0; j < a.Lk; ++j) {
                const float dsj = t.ds[wave][j], pdj = t.pd[wave][j];
                dq += dsj * t.k[j][lane];
                atomicAdd(&t.dk[j][lane], dsj * qv);
                atomicAdd(&t.dv[j][lane], pdj * gv);
            }
            st_elem(a.dq, qrow * a.lddq + h * a.dk + lane, a.dqkv_dtype, dq);
        }
        wave_sync();
    }
    __syncthreads();
    for (int idx = tid; idx < a.Lk * a.dk; idx += 256) {
        const int j = idx / a.dk, dd = idx - j * a.dk;
        const int64_t row = (int64_t)g * a.Lk + j;
        st_elem(a.d_k, row * a.lddk + h * a.dk + dd, a.dqkv_dtype, t.dk[j][dd]);
        st_elem(a.dv, row * a.lddv + h * a.dk + dd, a.dqkv_dtype, t.dv[j][dd]);
    }
}

int check(const ortk_attn_args* a) {
    if (!a || !a->q || !a->k || !a->v) return ORTK_EINVAL;
    if (a->Lk < 1 || a->Lk > MAXK || a->dk < 1 || a->dk > MAXD || a->Lq < 1 || a->H < 1 || a->nkv < 0) return ORTK_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return ORTK_EINVAL;
    return 0;
}

// the register-only kernels: training-time short sequences (decode steps with 1-8 query rows keep their own kernels)
bool small_ok(const ortk_attn_args* a) {
    return a->dk == 64 && a->Lq <= 32 && a->Lk <= 32 && a->Lq >= 9 && !a->kv_index;
}
// forward only: also the decode-time cross-attention (1-8 beams of an image x 36 regions: up to 3 key tiles)
bool small_fwd_ok(const ortk_attn_args* a) {
    return a->dk == 64 && a->Lq <= 32 && a->Lk <= 48 && !a->kv_index && (a->Lq >= 9 || a->Lk > 8);
}

}  // namespace

// cache append for the kernels that do not do it themselves (k_new / v_new): one thread per element of the new rows
__global__ void attn_kv_append_kernel(ortk_attn_args a) {
    const int d = a.H * a.dk;
    const int64_t n = (int64_t)a.nkv * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / d; const int c = (int)(i - g * d);
        const int j = a.Lk - 1;
        const int64_t row = a.kv_index ? (int64_t)a.kv_index[g * a.Lk + j] : g * (a.kv_group_stride > 0 ? a.kv_group_stride : a.Lk) + j;
        st_elem(const_cast<float*>(a.k), row * a.ldk + c, a.kv_dtype, a.k_new[g * a.ld_new + c]);
        st_elem(const_cast<float*>(a.v), row * a.ldv + c, a.kv_dtype, a.v_new[g * a.ld_new + c]);
    }
}

// bf16 K / V storage (kv_dtype = 1) is served by the two decode-time kernels only; ortk_attention_kv16_ok tells the
// executor whether a shape qualifies before it lays its caches out as bf16
static bool kv16_ok(const ortk_attn_args* a) {
    const bool rowdec = a->Lq == 1 && a->H == 8 && a->dk == 64 && a->Lk <= 32 && a->causal_period == 0 && a->drop_p == 0.f && !a->bias;
    const bool small = a->dk == 64 && a->Lq <= 16 && a->Lk <= 48 && !a->kv_index && (a->Lq >= 9 || a->Lk > 8) && !rowdec;
    return (rowdec || small) && a->ldk % 8 == 0 && a->ldv % 8 == 0 && attn_impl() == 0;
}

extern "C" int ortk_attention_fwd(const ortk_attn_args* a_in, ortk_stream stream) {
    if (int e = check(a_in)) return e;
    if (!a_in->o) return ORTK_EINVAL;
    if (a_in->nkv == 0) return 0;
    if (a_in->kv_dtype != 0 && (a_in->kv_dtype != 1 || !kv16_ok(a_in))) return ORTK_EINVAL;
    ortk_attn_args a_local = *a_in;
    ortk_attn_args* a = &a_local;
    if ((a->k_new != nullptr) != (a->v_new != nullptr)) return ORTK_EINVAL;
    if (a->k_new && (a->Lq != 1 || a->ld_new < (int64_t)a->H * a->dk)) return ORTK_EINVAL;
    if (a->k_new) {
        const bool rowdec = a->H == 8 && a->dk == 64 && a->Lk <= 32 && a->causal_period == 0 && a->drop_p == 0.f && !a->bias &&
                            a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->ldo % 4 == 0 && a->ld_new % 4 == 0 && attn_impl() == 0 &&
                            ((reinterpret_cast<uintptr_t>(a->q) | reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->v) |
                              reinterpret_cast<uintptr_t>(a->o) | reinterpret_cast<uintptr_t>(a->k_new) | reinterpret_cast<uintptr_t>(a->v_new)) & 15) == 0;
        if (!rowdec) {       // every other kernel: append first, then attend to the cache as usual
            const int64_t n = (int64_t)a->nkv * a->H * a->dk;
            hipLaunchKernelGGL(attn_kv_append_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ortk_cdiv(n, 256), 2048))), dim3(256), 0,
                               ortk_s(stream), *a);
            ORTK_CHECK_LAUNCH();
            a->k_new = a->v_new = nullptr;
        }
    }
    if ((attn_impl() == 0 || a->qkv_dtype) && ortk::attn16_ok(a, false)) return ortk::attn16_fwd(a, ortk_s(stream));   // mixed precision
    if (a->qkv_dtype || a->q_off) return ORTK_EINVAL;   // bf16 Q / K / V and ragged groups are only understood by the bf16-operand kernels
    ortk::lds_attr(reinterpret_cast<const void*>(attn_fwd_kernel), fwd_lds_bytes(MAXK));
    ortk::lds_attr(reinterpret_cast<const void*>(attn_fwd_wave_kernel), sizeof(float) * 4 * (2 * 64 * KP + 128));
    if (a->drop_tf_T > 0) {      // teacher-forced dropout geometry (train-mode decode step): the generic kernel
        if (a->kv_dtype || a->qkv_dtype || a->Lk > MAXK || a->dk > MAXD) return ORTK_EINVAL;
        hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(a->nkv * a->H)), dim3(256), fwd_lds_bytes(a->Lk), ortk_s(stream), *a);
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    // Dispatch measured on the path's three shapes (scratch/attn_bench.py, us fwd/bwd): 36x36 MFMA 81/132 vs 82/147,
    // 85x36 MFMA 101/218 vs 170/293, 17x17 (10 240 tiny pairs) MFMA 200/526 vs 138/196 for the per-lane kernels.
    const bool use_mfma = attn_impl() == 0 ? a->Lq > 32 : attn_impl() == 3;
    const bool vec_kq = (a->ldk % 4 == 0) && (a->ldq % 4 == 0) && (a->dk % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->q)) & 15) == 0;
    const bool al16 = ((reinterpret_cast<uintptr_t>(a->q) | reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->v) |
                        reinterpret_cast<uintptr_t>(a->o)) & 15) == 0;
    if (a->kv_dtype && !al16) return ORTK_EINVAL;
    if (a->Lq == 1 && a->H == 8 && a->dk == 64 && a->Lk <= 32 && a->causal_period == 0 && a->drop_p == 0.f && !a->bias && al16 &&
        a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->ldo % 4 == 0 && attn_impl() == 0) {
        const dim3 rgrid((unsigned)ortk_cdiv(a->nkv, 4));
#define ORTK_ROWDEC(LK) do { if (a->kv_dtype) hipLaunchKernelGGL((attn_rowdec_kernel<LK, __bf16>), rgrid, dim3(256), 0, ortk_s(stream), *a); \
                             else             hipLaunchKernelGGL((attn_rowdec_kernel<LK, float>), rgrid, dim3(256), 0, ortk_s(stream), *a); } while (0)
        if (a->Lk <= 8) ORTK_ROWDEC(8); else if (a->Lk <= 16) ORTK_ROWDEC(16); else if (a->Lk <= 24) ORTK_ROWDEC(24); else ORTK_ROWDEC(32);
#undef ORTK_ROWDEC
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    // from here on only the register-only MFMA kernel understands bf16 K / V
    if (a->kv_dtype && !(small_fwd_ok(a) && attn_impl() == 0 && a->ldv % 8 == 0 && a->ldq % 4 == 0 && a->ldk % 8 == 0 && al16))
        return ORTK_EINVAL;
    // measured on the 1024-image beam-5 decode: 95-105 us per call vs 75 us for the block kernel -> opt-in only (impl 4)
    if (a->Lq <= 8 && a->Lk <= 64 && a->causal_period == 0 && a->drop_p == 0.f && vec_kq && attn_impl() == 4) {
        const dim3 dgrid((unsigned)ortk_cdiv((int64_t)a->nkv * a->H, 4));
        if (a->Lk <= 32) hipLaunchKernelGGL(attn_decode_kernel<32>, dgrid, dim3(256), 0, ortk_s(stream), *a);
        else             hipLaunchKernelGGL(attn_decode_kernel<64>, dgrid, dim3(256), 0, ortk_s(stream), *a);
    } else if (small_fwd_ok(a) && attn_impl() == 0 && a->ldv % 4 == 0 && a->ldq % 4 == 0 && a->ldk % 4 == 0 &&
               ((reinterpret_cast<uintptr_t>(a->q) | reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->v)) & 15) == 0) {
        const dim3 sgrid((unsigned)ortk_cdiv((int64_t)a->nkv * a->H, 4));
        const int nit = a->Lq > 16 ? 2 : 1, njt = (a->Lk + 15) / 16;
#define ORTK_SMALL_FWD(NI, NJ, KT) hipLaunchKernelGGL((attn_small_fwd_kernel<NI, NJ, KT>), sgrid, dim3(256), 0, ortk_s(stream), *a)
        if (a->kv_dtype) {   // bf16 K / V: decode-time shapes (at most 16 query rows)
            if (nit != 1) return ORTK_EINVAL;
            if (njt == 3) ORTK_SMALL_FWD(1, 3, __bf16); else if (njt == 2) ORTK_SMALL_FWD(1, 2, __bf16); else ORTK_SMALL_FWD(1, 1, __bf16);
        } else if (nit == 2) { if (njt == 3) ORTK_SMALL_FWD(2, 3, float); else if (njt == 2) ORTK_SMALL_FWD(2, 2, float); else ORTK_SMALL_FWD(2, 1, float); }
        else                 { if (njt == 3) ORTK_SMALL_FWD(1, 3, float); else if (njt == 2) ORTK_SMALL_FWD(1, 2, float); else ORTK_SMALL_FWD(1, 1, float); }
#undef ORTK_SMALL_FWD
    } else if (a->Lk <= 64 && use_mfma && vec_kq && a->ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(a->v) & 15) == 0) {
        // MFMA form: one workgroup per pair, one wave per 16 query rows (at most 8 waves)
        const int Lkp = (int)ortk_align(a->Lk, 16), DKP = (int)ortk_align(a->dk, 16);
        const int nw = (int)std::min<int64_t>(8, ortk_cdiv(a->Lq, 16));
        const size_t lds = sizeof(float) * ((size_t)Lkp * (PK_ + PN_) + 64 + (size_t)nw * 16 * PK_ * 2);
        ortk::lds_attr(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<0, 0>), 160 * 1024);
        ortk::lds_attr(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<64, 64>), 160 * 1024);
        const dim3 mgrid((unsigned)(a->nkv * a->H)), mblock(64 * nw);
        if (DKP == 64 && Lkp == 48)      hipLaunchKernelGGL((attn_fwd_mfma_kernel<48, 64>), mgrid, mblock, lds, ortk_s(stream), *a, Lkp, DKP);
        else if (DKP == 64 && Lkp == 32) hipLaunchKernelGGL((attn_fwd_mfma_kernel<32, 64>), mgrid, mblock, lds, ortk_s(stream), *a, Lkp, DKP);
        else if (DKP == 64 && Lkp == 64) hipLaunchKernelGGL((attn_fwd_mfma_kernel<64, 64>), mgrid, mblock, lds, ortk_s(stream), *a, Lkp, DKP);
        else                             hipLaunchKernelGGL((attn_fwd_mfma_kernel<0, 0>), mgrid, mblock, lds, ortk_s(stream), *a, Lkp, DKP);
    } else if (a->Lk <= 64 && attn_impl() == 1) {
        const int pairs = a->nkv * a->H;
        hipLaunchKernelGGL(attn_fwd_wave_kernel, dim3((unsigned)ortk_cdiv(pairs, 4)), dim3(256),
                           sizeof(float) * 4 * ((size_t)2 * a->Lk * KP + 128), ortk_s(stream), *a);
    } else {
        hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(a->nkv * a->H)), dim3(256), fwd_lds_bytes(a->Lk), ortk_s(stream), *a);
    }
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_attention_bwd(const ortk_attn_args* a, ortk_stream stream) {
    if (int e = check(a)) return e;
    if (a->kv_dtype != 0 || a->bwd_part < 0 || a->bwd_part > 2) return ORTK_EINVAL;
    if (!a->p || !a->d_o || !a->dq || !a->d_k || !a->dv || a->kv_index || a->kv_group_stride) return ORTK_EINVAL;
    if (a->nkv == 0) return 0;
    if ((attn_impl() == 0 || a->qkv_dtype) && ortk::attn16_ok(a, true))     // mixed precision: one kernel does both parts
        return a->bwd_part == 2 ? 0 : ortk::attn16_bwd(a, ortk_s(stream));
    if (a->qkv_dtype || a->q_off) return ORTK_EINVAL;
    ortk::lds_attr(reinterpret_cast<const void*>(attn_bwd_kernel), bwd_lds_bytes(MAXK));
    ortk::lds_attr(reinterpret_cast<const void*>(attn_bwd_wave_kernel<64>), sizeof(float) * 4 * (2 * 64 * KP + 192));
    ortk::lds_attr(reinterpret_cast<const void*>(attn_bwd_wave_kernel<48>), sizeof(float) * 4 * (2 * 48 * KP + 192));
    const int pairs = a->nkv * a->H;
    if (small_ok(a) && attn_impl() == 0 && a->ldv % 4 == 0 && a->lddo % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(a->v) | reinterpret_cast<uintptr_t>(a->d_o)) & 15) == 0) {
        const dim3 sgrid((unsigned)ortk_cdiv(pairs, 4));
        const int nit = a->Lq > 16 ? 2 : 1, njt = a->Lk > 16 ? 2 : 1;
        if (nit == 2 && njt == 2)      hipLaunchKernelGGL((attn_small_bwd_kernel<2, 2>), sgrid, dim3(256), 0, ortk_s(stream), *a);
        else if (nit == 2)             hipLaunchKernelGGL((attn_small_bwd_kernel<2, 1>), sgrid, dim3(256), 0, ortk_s(stream), *a);
        else if (njt == 2)             hipLaunchKernelGGL((attn_small_bwd_kernel<1, 2>), sgrid, dim3(256), 0, ortk_s(stream), *a);
        else                           hipLaunchKernelGGL((attn_small_bwd_kernel<1, 1>), sgrid, dim3(256), 0, ortk_s(stream), *a);
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    {
        const int Lkp = (int)ortk_align(a->Lk, 16), DKP = (int)ortk_align(a->dk, 16), Lqp = (int)ortk_align(a->Lq, 16);
        const size_t lds = sizeof(float) * ((size_t)Lkp * (PK_ + PN_) + (size_t)Lqp * (PN_ + 3 * PK_));
        const bool use_mfma = attn_impl() == 0 ? a->Lq > 32 : attn_impl() == 3;
        const bool vec = a->dk % 4 == 0 && a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->lddo % 4 == 0 &&
                         ((reinterpret_cast<uintptr_t>(a->q) | reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->v) |
                           reinterpret_cast<uintptr_t>(a->d_o)) & 15) == 0;
        if (a->Lk <= 64 && vec && lds <= 160 * 1024 && use_mfma) {
            const int nw = (int)std::min<int64_t>(8, ortk_cdiv(a->Lq, 16));
            const dim3 mgrid((unsigned)pairs), mblock(64 * nw);
            const size_t lds1 = sizeof(float) * ((size_t)Lkp * (PN_ + PK_) + (size_t)Lqp * 2 * PK_);    // PART 1: no Q / dropped-P images
            typedef void (*bwd_fn)(ortk_attn_args, int, int, int);
            bwd_fn fn; size_t bytes = lds;
            if (a->bwd_part == 1 && DKP == 64 && Lkp == 48 && Lqp == 96)      { fn = attn_bwd_mfma_kernel<48, 64, 96, 1>; bytes = lds1; }
            else if (a->bwd_part == 2 && DKP == 64 && Lkp == 48 && Lqp == 96) fn = attn_bwd_mfma_kernel<48, 64, 96, 2>;
            else if (a->bwd_part == 1)                                        { fn = attn_bwd_mfma_kernel<0, 0, 0, 1>; bytes = lds1; }
            else if (a->bwd_part == 2)                                        fn = attn_bwd_mfma_kernel<0, 0, 0, 2>;
            else if (DKP == 64 && Lkp == 48 && Lqp == 48)                     fn = attn_bwd_mfma_kernel<48, 64, 48, 0>;
            else if (DKP == 64 && Lkp == 48 && Lqp == 96)                     fn = attn_bwd_mfma_kernel<48, 64, 96, 0>;
            else if (DKP == 64 && Lkp == 32 && Lqp == 32)                     fn = attn_bwd_mfma_kernel<32, 64, 32, 0>;
            else                                                              fn = attn_bwd_mfma_kernel<0, 0, 0, 0>;
            ortk::lds_attr(reinterpret_cast<const void*>(fn), 160 * 1024);
            hipLaunchKernelGGL(fn, mgrid, mblock, bytes, ortk_s(stream), *a, Lkp, DKP, Lqp);
            ORTK_CHECK_LAUNCH();
            return 0;
        }
    }
    if (a->bwd_part == 2) return 0;     // the kernels below compute everything in one go: part 1 already did
    const dim3 wgrid((unsigned)ortk_cdiv(pairs, 4)), block(256);
    const size_t wave_lds = sizeof(float) * 4 * ((size_t)2 * a->Lk * KP + 192);
    if (a->Lk <= 32)
        hipLaunchKernelGGL(attn_bwd_wave_kernel<32>, wgrid, block, wave_lds, ortk_s(stream), *a);
    else if (a->Lk <= 48)
        hipLaunchKernelGGL(attn_bwd_wave_kernel<48>, wgrid, block, wave_lds, ortk_s(stream), *a);
    else if (a->Lk <= 64)
        hipLaunchKernelGGL(attn_bwd_wave_kernel<64>, wgrid, block, wave_lds, ortk_s(stream), *a);
    else
        hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)pairs), block, bwd_lds_bytes(a->Lk), ortk_s(stream), *a);
    ORTK_CHECK_LAUNCH();
    return 0;
}
