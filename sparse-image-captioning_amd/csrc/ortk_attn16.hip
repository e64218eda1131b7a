// ortk_attn16.hip — bf16-MFMA attention for the mixed-precision mode (ortk_attn_args.precision = 1), forward and backward,
// for the two block shapes of the training step: encoder self-attention (36 x 36 regions, additive geometry bias) and
// decoder cross-attention (all captions of an image, 5 x 17 = 85 query rows, against its 36 regions); dk = 64.
//
// Same mathematics as the fp32-MFMA kernels of ortk_attn.hip (transformer.py:285-295, relation_transformer.py:258-293)
// — score = q.k / sqrt(dk); score[mask == 0] = -1e9; score += bias; P = softmax; O = dropout(P) V — with the operands of
// the four small products rounded to bf16 (fp32 accumulation, fp32 soft-max, fp32 P saved for the backward).  Why a
// second family: the fp32 kernels keep fp32 images of Q, K, V, dO, dS, P in LDS (109 KB for the cross-attention backward:
// ONE workgroup per CU, its load -> compute -> store phases fully exposed, 166 us where the HBM traffic needs 40) and
// feed v_mfma_f32_16x16x4_f32 one dword per lane and multiply-add.  Here every image is bf16 with 144-byte rows (30 KB
// forward, 53-71 KB backward: 2-5 workgroups per CU), a fragment is one ds_read_b128 or two ds_read_b64_tr_b16, and a
// 16 x 16 x 64 product is two v_mfma_f32_16x16x32_bf16.
//
// Layout rules (lane = (lr = lane & 15, lq = lane >> 4); MFMA(X, Y) gives D[x = 4*lq + r][y = lr], both operands
// supplying [index = lr][k = 8*lq .. 8*lq + 7]):
//   * every product is computed TRANSPOSED so that a lane ends up with 4 CONSECUTIVE elements of an output row:
//     scores S^T[key j = 16*jt + 4*lq + r][query i = lr] (soft-max over keys = in-lane + 2 shuffles; P rows, bias rows and
//     the LDS images of P / dS are 16- and 8-byte vector accesses), outputs O / dQ / dK / dV as [feature 4 consecutive][row];
//   * an operand whose k index runs along its image ROWS (K in dS.K, Q / dO / dS / P in the dK, dV products, V in P.V) is
//     gathered by the hardware-transposing LDS read; padded k ranges (keys 36 -> 64, queries 85 -> 96, 36 -> 64) are zero
//     rows / columns of the images.
// The pitch of 72 bf16 (144 B) makes the 16 rows of a ds_read_b128 service group fall on 16 distinct 16-byte slots.
#include <cstdlib>
#include "ortk_internal.h"

namespace {

constexpr int P16 = 72;                 // bf16 elements per row of an image with up to 64 columns (keys, or the 64 features of a head)

__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Key-indexed geometry for NJT key tiles (Lk <= 16 * NJT <= 128): KJ = key range as a k dimension (multiple of 32, zero-
// padded), PJ = pitch of the [query][key] images (KJ + 8: the same 16-distinct-slots property as P16 for its row length).
template <int NJT> struct KeyGeo {
    static constexpr int KJ = NJT <= 2 ? 32 : NJT <= 4 ? 64 : NJT <= 6 ? 96 : 128;
    static constexpr int PJ = NJT <= 4 ? P16 : 136;
};

// [index = row][k = k0 .. k0+7]: 8 consecutive elements of an image row
template <int PITCH = P16>
__device__ __forceinline__ bf16x8 frag_row(const __bf16* img, int row, int k0) {
    return *reinterpret_cast<const bf16x8*>(img + row * PITCH + k0);
}
// [index = x0 + lr][k = k0 + 8*lq .. +7] where k runs along the image ROWS: two transposing reads of 4 rows x 16 columns
template <int PITCH = P16>
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, int k0, int x0, int lane) {
    const int lr = lane & 15, lq = lane >> 4;
    const __bf16* p = img + (k0 + 8 * lq + (lr >> 2)) * PITCH + x0 + 4 * (lane & 3);
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p + 4 * PITCH));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x4 cvt4(float4 v) { return (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; }
__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b) {
    return (bf16x8){(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}
__device__ __forceinline__ f32x4 mma(bf16x8 x, bf16x8 y, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0); }

// rows [0, rows) of a (.., 64)-column matrix slice (fp32 or bf16 in memory) -> bf16 image; rows [rows, rows_pad) are zeroed
// (DKT = features per head: 64 or 32; image pitch DKT + 8)
template <int DKT>
__device__ __forceinline__ void stage_rows(__bf16* img, const float* src, int64_t ld, int rows, int rows_pad, int tid, int nthr) {
    constexpr int CPR = DKT / 4;
    for (int idx = tid; idx < rows_pad * CPR; idx += nthr) {
        const int r = idx / CPR, c = (idx % CPR) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c);
        *reinterpret_cast<bf16x4*>(img + r * (DKT + 8) + c) = cvt4(v);
    }
}
template <int DKT>
__device__ __forceinline__ void stage_rows(__bf16* img, const __bf16* src, int64_t ld, int rows, int rows_pad, int tid, int nthr) {
    constexpr int CPR = DKT / 8;
    for (int idx = tid; idx < rows_pad * CPR; idx += nthr) {
        const int r = idx / CPR, c = (idx % CPR) * 8;
        bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (r < rows) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * ld + c);
        *reinterpret_cast<bf16x8*>(img + r * (DKT + 8) + c) = v;
    }
}
// 8 consecutive elements of a global row as an MFMA fragment
__device__ __forceinline__ bf16x8 ld_frag(const float* p) { return cvt8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)); }
__device__ __forceinline__ bf16x8 ld_frag(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// ------------------------------------------------------------------------------------------------ forward
// One workgroup per (group g, head h); one wave per 16-row query tile.  NJT = key tiles (Lk <= 16 * NJT <= 128).
template <int NJT, typename TQ, int DKT>
__global__ __launch_bounds__(512) void attn16_fwd_kernel(ortk_attn_args a) {
    constexpr int DK = DKT, PD = DKT + 8;           // features per head; pitch of the [row][feature] images
    const float qscale = DKT == 64 ? 0.125f : 0.17677669529663687f;      // 1 / sqrt(dk)
    const TQ* aq = reinterpret_cast<const TQ*>(a.q); const TQ* ak = reinterpret_cast<const TQ*>(a.k); const TQ* av = reinterpret_cast<const TQ*>(a.v);
    extern __shared__ __attribute__((aligned(16))) __bf16 sm16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int Lk = a.Lk, Lq = a.Lq;
    constexpr int KJ = KeyGeo<NJT>::KJ, PJ = KeyGeo<NJT>::PJ;
    __bf16* sK = sm16;                              // [16*NJT][PD]   rows = key            (row fragments)
    __bf16* sV = sK + 16 * NJT * PD;                // [KJ][PD]       rows = key, zero-padded (k of P.V: transposing reads)
    __bf16* sP = sV + KJ * PD + wave * 16 * PJ;     // [16][PJ]       this wave's dropped P: rows = query, columns = key 0..KJ-1
    float* sMask = reinterpret_cast<float*>(sm16 + (16 * NJT + KJ) * PD + 16 * nw * PJ);   // [128]
    const int lr = lane & 15, lq = lane >> 4;
    const bool vec_p = (Lk & 3) == 0;               // P / bias rows start 16-byte aligned
    const int nit = (Lq + 15) >> 4;
    // ragged groups (ortk_attn_args.q_off): the group's first query row and its row count; with kv_ragged its keys are those rows
    const int64_t qrow0 = a.q_off ? (int64_t)a.q_off[(int64_t)g * a.q_off_stride] : (int64_t)g * Lq;
    const int Lqg = a.q_off ? min(Lq, (int)(a.q_off[(int64_t)(g + 1) * a.q_off_stride] - qrow0)) : Lq;
    const int64_t krow0 = (a.q_off && a.kv_ragged) ? qrow0 : (int64_t)g * Lk;
    const int Lkg = (a.q_off && a.kv_ragged) ? min(Lk, Lqg) : Lk;
    // The launcher gives every 16-row query tile its own wave (nw = nit).  This wave's Q fragments and bias rows are
    // requested BEFORE the K / V staging and its barrier, so that the workgroup pays one global round trip, not two.
    const int it = wave;
    const int i = it * 16 + lr;                     // this lane's query (operand row and output row)
    const bool iv = it < nit && i < Lqg;
    const int64_t prow = (((int64_t)g * a.H + h) * Lq + i) * Lk;
    // dropout draws of a ragged group keyed like the unragged layout's (ortk_attn_args.drop_rows): query i -> drop_rows[.] - g Lq
    const int64_t drow = (a.drop_rows && a.q_off && iv && a.drop_p > 0.f)
                             ? (((int64_t)g * a.H + h) * Lq + ((int64_t)a.drop_rows[qrow0 + i] - (int64_t)g * Lq)) * Lk : prow;
    bf16x8 qf[DK / 32];
#pragma unroll
    for (int ks = 0; ks < DK / 32; ++ks) qf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (iv) {
        const TQ* qp = aq + (qrow0 + i) * a.ldq + h * DK + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < DK / 32; ++ks) qf[ks] = ld_frag(qp + 32 * ks);
    }
    float4 bias4[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int j0 = 16 * jt + 4 * lq;
        bias4[jt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && iv) {
            if (vec_p && j0 < Lk) bias4[jt] = *reinterpret_cast<const float4*>(a.bias + prow + j0);
            else { float* bp = &bias4[jt].x; for (int r = 0; r < 4; ++r) if (j0 + r < Lk) bp[r] = a.bias[prow + j0 + r]; }
        }
    }
    stage_rows<DK>(sK, ak + krow0 * a.ldk + h * DK, a.ldk, Lkg, 16 * NJT, tid, blockDim.x);
    stage_rows<DK>(sV, av + krow0 * a.ldv + h * DK, a.ldv, Lkg, KJ, tid, blockDim.x);
    if (tid < 128) sMask[tid] = (tid < Lkg) ? (a.kmask ? a.kmask[krow0 + tid] : 1.f) : -1.f;   // -1: padded key
    // key columns 16*NJT .. KJ-1 of the P image are never written below: zero the image once
    for (int idx = lane; idx < 16 * (KJ / 4); idx += 64)
        *reinterpret_cast<bf16x4*>(sP + (idx / (KJ / 4)) * PJ + (idx % (KJ / 4)) * 4) = cvt4(make_float4(0.f, 0.f, 0.f, 0.f));
    __syncthreads();
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    if (it < nit) {
        // S^T[j = 16*jt + 4*lq + r][i = lr]
        f32x4 s[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DK / 32; ++ks) s[jt] = mma(frag_row<PD>(sK, 16 * jt + lr, 32 * ks + 8 * lq), qf[ks], s[jt]);
        }
        const int qpos = a.causal_period > 0 ? i % a.causal_period : 0;
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int j0 = 16 * jt + 4 * lq;
            const float bb[4] = {bias4[jt].x, bias4[jt].y, bias4[jt].z, bias4[jt].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + r;
                const float mk = sMask[j];
                float x = s[jt][r] * qscale;
                if (mk == 0.f || (a.causal_period > 0 && j > qpos)) x = -1e9f;
                if (mk < 0.f) x = -INFINITY;                                      // padded key: not part of the row
                else if (a.bias && iv) x = bb[r] + x;
                s[jt][r] = x;
                mx = fmaxf(mx, x);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = expf(s[jt][r] - mx); s[jt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int j0 = 16 * jt + 4 * lq;
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = (iv && j0 + r < Lk) ? s[jt][r] / sum : 0.f;
            if (a.p && iv) {
                if (vec_p && j0 < Lk) *reinterpret_cast<float4*>(a.p + prow + j0) = make_float4(p[0], p[1], p[2], p[3]);
                else for (int r = 0; r < 4; ++r) if (j0 + r < Lk) a.p[prow + j0 + r] = p[r];
            }
            if (a.drop_p > 0.f) {
                bool kp[4];
                ortk_keep4(a.drop_seed, (uint64_t)(drow + j0), a.drop_p, kp);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = kp[r] ? p[r] * inv_keep : 0.f;
            }
            *reinterpret_cast<bf16x4*>(sP + lr * PJ + j0) = cvt4(make_float4(p[0], p[1], p[2], p[3]));
        }
        wsync();
        // O^T[d = 16*dt + 4*lq + r][i = lr] = sum_j V[j][d] Pd[i][j]
        bf16x8 pf[KJ / 32];
#pragma unroll
        for (int ks = 0; ks < KJ / 32; ++ks) pf[ks] = frag_row<PJ>(sP, lr, 32 * ks + 8 * lq);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KJ / 32; ++ks) o = mma(frag_tr<PD>(sV, 32 * ks, 16 * dt, lane), pf[ks], o);
            if (iv) st_elem4(a.o, (qrow0 + i) * a.ldo + h * DK + 16 * dt + 4 * lq, a.o_dtype, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}
static int key_kj(int njt) { return njt <= 2 ? 32 : njt <= 4 ? 64 : njt <= 6 ? 96 : 128; }
static int key_pj(int njt) { return njt <= 4 ? P16 : 136; }
static size_t fwd16_lds(int njt, int nw, int dk) {
    return ((size_t)(16 * njt + key_kj(njt)) * (dk + 8) + (size_t)16 * nw * key_pj(njt)) * sizeof(__bf16) + 128 * sizeof(float);
}

// ------------------------------------------------------------------------------------------------ backward
// Phase 1 (wave = 16-row query tile): dP^T = V dO^T, dS = P (dP - rowsum(P dP)), dQ = dS K / sqrt(dk); dS / sqrt(dk) and the
// dropped P go to workgroup-wide images.  Phase 2 (the 2 x NJT x 4 output tiles shared by the waves): dK = dS^T Q / sqrt(dk),
// dV = Pd^T dO over all query rows of the group.  Lqp = query rows padded to a multiple of 32 (k range of phase 2).
template <int NJT, typename TQ, int DKT>
__global__ __launch_bounds__(512) void attn16_bwd_kernel(ortk_attn_args a, int Lqp) {
    constexpr int DK = DKT, PD = DKT + 8;
    const float qscale = DKT == 64 ? 0.125f : 0.17677669529663687f;
    const TQ* aq = reinterpret_cast<const TQ*>(a.q); const TQ* ak = reinterpret_cast<const TQ*>(a.k); const TQ* av = reinterpret_cast<const TQ*>(a.v);
    extern __shared__ __attribute__((aligned(16))) __bf16 sm16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x / a.H, h = blockIdx.x - g * a.H;
    const int Lk = a.Lk, Lq = a.Lq;
    constexpr int KJ = KeyGeo<NJT>::KJ, PJ = KeyGeo<NJT>::PJ;
    __bf16* sK = sm16;                      // [KJ][PD]       rows = key, zero-padded   (k of dS.K: transposing reads)
    __bf16* sV = sK + KJ * PD;              // [16*NJT][PD]   rows = key                (row fragments of dP^T = V dO^T)
    __bf16* sQ = sV + 16 * NJT * PD;        // [Lqp][PD]      rows = query              (k of dS^T Q: transposing reads)
    __bf16* sG = sQ + Lqp * PD;             // [Lqp][PD]      dO: row fragments in phase 1, transposing reads in phase 2
    __bf16* sS = sG + Lqp * PD;             // [Lqp][PJ]      dS / sqrt(dk): rows = query, columns = key 0..KJ-1
    __bf16* sD = sS + Lqp * PJ;             // [Lqp][PJ]      dropped P
    const int lr = lane & 15, lq = lane >> 4;
    const bool vec_p = (Lk & 3) == 0;
    const int nit = (Lq + 15) >> 4;
    // ragged groups: see the forward kernel.  Query rows past the group's count are staged as ZERO rows of Q and dO (they are
    // another group's rows): they add nothing to dK / dV
    const int64_t qrow0 = a.q_off ? (int64_t)a.q_off[(int64_t)g * a.q_off_stride] : (int64_t)g * Lq;
    const int Lqg = a.q_off ? min(Lq, (int)(a.q_off[(int64_t)(g + 1) * a.q_off_stride] - qrow0)) : Lq;
    const int64_t krow0 = (a.q_off && a.kv_ragged) ? qrow0 : (int64_t)g * Lk;
    const int Lkg = (a.q_off && a.kv_ragged) ? min(Lk, Lqg) : Lk;
    // one 16-row query tile per wave (nw = nit); its saved probabilities are requested before the staging barrier
    const int it = wave;
    const int i0 = it * 16, i = i0 + lr;
    const bool iv = it < nit && i < Lqg;
    const int64_t prow = (((int64_t)g * a.H + h) * Lq + i) * Lk;
    const int64_t drow = (a.drop_rows && a.q_off && iv && a.drop_p > 0.f)
                             ? (((int64_t)g * a.H + h) * Lq + ((int64_t)a.drop_rows[qrow0 + i] - (int64_t)g * Lq)) * Lk : prow;
    float4 praw[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int j0 = 16 * jt + 4 * lq;
        praw[jt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iv) {
            if (vec_p && j0 < Lk) praw[jt] = *reinterpret_cast<const float4*>(a.p + prow + j0);
            else { float* q = &praw[jt].x; for (int r = 0; r < 4; ++r) if (j0 + r < Lk) q[r] = a.p[prow + j0 + r]; }
        }
    }
    stage_rows<DK>(sK, ak + krow0 * a.ldk + h * DK, a.ldk, Lkg, KJ, tid, blockDim.x);
    stage_rows<DK>(sV, av + krow0 * a.ldv + h * DK, a.ldv, Lkg, 16 * NJT, tid, blockDim.x);
    stage_rows<DK>(sQ, aq + qrow0 * a.ldq + h * DK, a.ldq, Lqg, Lqp, tid, blockDim.x);
    stage_rows<DK>(sG, reinterpret_cast<const TQ*>(a.d_o) + qrow0 * a.lddo + h * DK, a.lddo, Lqg, Lqp, tid, blockDim.x);
    // dS / P images: the key columns 16*NJT .. 63 and the query rows past the last wave tile are never written below
    for (int idx = tid; idx < 2 * Lqp * (KJ / 4); idx += blockDim.x)
        *reinterpret_cast<bf16x4*>(sS + (idx / (KJ / 4)) * PJ + (idx % (KJ / 4)) * 4) = cvt4(make_float4(0.f, 0.f, 0.f, 0.f));   // sD follows sS
    __syncthreads();
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    if (it < nit) {
        bf16x8 gf[DK / 32];
#pragma unroll
        for (int ks = 0; ks < DK / 32; ++ks) gf[ks] = frag_row<PD>(sG, i, 32 * ks + 8 * lq);
        // dP^T[j = 16*jt + 4*lq + r][i = lr]
        f32x4 dp[NJT], pp[NJT];
        float dot = 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            dp[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DK / 32; ++ks) dp[jt] = mma(frag_row<PD>(sV, 16 * jt + lr, 32 * ks + 8 * lq), gf[ks], dp[jt]);
            const int j0 = 16 * jt + 4 * lq;
            const float pv[4] = {praw[jt].x, praw[jt].y, praw[jt].z, praw[jt].w};
            float pd[4];
            bool kp[4] = {true, true, true, true};
            if (a.drop_p > 0.f) ortk_keep4(a.drop_seed, (uint64_t)(drow + j0), a.drop_p, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = iv && j0 + r < Lk;
                const bool keep = valid && kp[r];
                const float d = keep ? dp[jt][r] * inv_keep : 0.f;
                pd[r] = keep ? pv[r] * inv_keep : 0.f;
                pp[jt][r] = valid ? pv[r] : 0.f; dp[jt][r] = d; dot += pp[jt][r] * d;
            }
            *reinterpret_cast<bf16x4*>(sD + i * PJ + j0) = cvt4(make_float4(pd[0], pd[1], pd[2], pd[3]));
        }
        dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int j0 = 16 * jt + 4 * lq;
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] = pp[jt][r] * (dp[jt][r] - dot);
            if (a.dscore && iv) {
                if (vec_p && j0 < Lk) *reinterpret_cast<float4*>(a.dscore + prow + j0) = make_float4(ds[0], ds[1], ds[2], ds[3]);
                else for (int r = 0; r < 4; ++r) if (j0 + r < Lk) a.dscore[prow + j0 + r] = ds[r];
            }
            *reinterpret_cast<bf16x4*>(sS + i * PJ + j0) = cvt4(make_float4(ds[0] * qscale, ds[1] * qscale, ds[2] * qscale, ds[3] * qscale));
        }
        wsync();
        // dQ^T[d = 16*dt + 4*lq + r][i = lr] = sum_j K[j][d] dSs[i][j]
        bf16x8 sf[KJ / 32];
#pragma unroll
        for (int ks = 0; ks < KJ / 32; ++ks) sf[ks] = frag_row<PJ>(sS, i, 32 * ks + 8 * lq);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KJ / 32; ++ks) acc = mma(frag_tr<PD>(sK, 32 * ks, 16 * dt, lane), sf[ks], acc);
            if (iv) st_elem4(a.dq, (qrow0 + i) * a.lddq + h * DK + 16 * dt + 4 * lq, a.dqkv_dtype, make_float4(acc[0], acc[1], acc[2], acc[3]));
        }
    }
    __syncthreads();
    // phase 2: tile t = (which, jt, dt): D[d = 16*dt + 4*lq + r][j = 16*jt + lr] = sum_i B[i][d] A[i][j]
    constexpr int NDT = DK / 16;
    const int ntiles = 2 * NJT * NDT;
    for (int t = wave; t < ntiles; t += nw) {
        const int which = t / (NJT * NDT), rem = t - which * NJT * NDT, jt = rem / NDT, dt = rem % NDT;
        const __bf16* sA = which == 0 ? sS : sD;      // [query][key]
        const __bf16* sB = which == 0 ? sQ : sG;      // [query][feature]
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < Lqp; k0 += 32) acc = mma(frag_tr<PD>(sB, k0, 16 * dt, lane), frag_tr<PJ>(sA, k0, 16 * jt, lane), acc);
        const int j = 16 * jt + lr;
        if (j < Lkg) {
            const int64_t row = krow0 + j;
            const float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (which == 0) st_elem4(a.d_k, row * a.lddk + h * DK + 16 * dt + 4 * lq, a.dqkv_dtype, v);
            else            st_elem4(a.dv, row * a.lddv + h * DK + 16 * dt + 4 * lq, a.dqkv_dtype, v);
        }
    }
}
static size_t bwd16_lds(int njt, int Lqp, int dk) {
    return ((size_t)(key_kj(njt) + 16 * njt + 2 * Lqp) * (dk + 8) + (size_t)2 * Lqp * key_pj(njt)) * sizeof(__bf16);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace ortk {

// shapes / layouts these kernels serve (everything else stays with the fp32-MFMA family)
bool attn16_shape_ok(int Lq, int Lk, int dk) { return (dk == 64 || dk == 32) && Lk >= 1 && Lk <= 128 && Lq >= 1 && Lq <= 128; }

bool attn16_ok(const ortk_attn_args* a, bool bwd) {
    // fp32 inputs: only the block shapes (the register-only kernels keep the short ones); bf16 inputs: every served shape
    const int min_lq = ortk::tuning().attn16_min_lq;
    // (short query blocks over at most 48 keys stay with the register-only kernels; past 48 keys those do not apply)
    if (a->precision != 1 || !attn16_shape_ok(a->Lq, a->Lk, a->dk) || (a->qkv_dtype == 0 && a->Lq < min_lq && a->Lk <= 48)) return false;
    if (a->kv_index || a->kv_group_stride > 0 || a->kv_dtype != 0 || a->k_new || a->v_new) return false;
    if (a->q_off && (a->qkv_dtype != 1 || a->q_off_stride < 1)) return false;
    if ((a->ldq | a->ldk | a->ldv) % (a->qkv_dtype ? 8 : 4) || !al16(a->q) || !al16(a->k) || !al16(a->v)) return false;
    if (!bwd) {
        const int64_t eo = a->o_dtype == ORTK_BF16 ? 2 : 4;
        if (a->ldo % 4 || (reinterpret_cast<uintptr_t>(a->o) % (4 * eo)) || (a->p && !al16(a->p)) || (a->bias && !al16(a->bias))) return false;
    } else {
        const int64_t eg = a->dqkv_dtype == ORTK_BF16 ? 2 : 4;
        if (!a->p || !a->d_o || !a->dq || !a->d_k || !a->dv) return false;
        if ((a->lddo % (a->qkv_dtype ? 8 : 4)) || (a->lddq | a->lddk | a->lddv) % 4 || !al16(a->d_o) || !al16(a->p) || (a->dscore && !al16(a->dscore))) return false;
        if ((reinterpret_cast<uintptr_t>(a->dq) | reinterpret_cast<uintptr_t>(a->d_k) | reinterpret_cast<uintptr_t>(a->dv)) % (4 * eg)) return false;
    }
    return true;
}

typedef void (*fwd16_fn)(ortk_attn_args);
typedef void (*bwd16_fn)(ortk_attn_args, int);
template <typename TQ, int DKT> static fwd16_fn pick_fwd(int njt) {
    switch (njt) {
        case 1: return attn16_fwd_kernel<1, TQ, DKT>; case 2: return attn16_fwd_kernel<2, TQ, DKT>; case 3: return attn16_fwd_kernel<3, TQ, DKT>;
        case 4: return attn16_fwd_kernel<4, TQ, DKT>; case 5: return attn16_fwd_kernel<5, TQ, DKT>; case 6: return attn16_fwd_kernel<6, TQ, DKT>;
        case 7: return attn16_fwd_kernel<7, TQ, DKT>; default: return attn16_fwd_kernel<8, TQ, DKT>;
    }
}
template <typename TQ, int DKT> static bwd16_fn pick_bwd(int njt) {
    switch (njt) {
        case 1: return attn16_bwd_kernel<1, TQ, DKT>; case 2: return attn16_bwd_kernel<2, TQ, DKT>; case 3: return attn16_bwd_kernel<3, TQ, DKT>;
        case 4: return attn16_bwd_kernel<4, TQ, DKT>; case 5: return attn16_bwd_kernel<5, TQ, DKT>; case 6: return attn16_bwd_kernel<6, TQ, DKT>;
        case 7: return attn16_bwd_kernel<7, TQ, DKT>; default: return attn16_bwd_kernel<8, TQ, DKT>;
    }
}

int attn16_fwd(const ortk_attn_args* a, hipStream_t s) {
    const int njt = (a->Lk + 15) / 16, nw = (a->Lq + 15) / 16;
    const fwd16_fn fn = a->dk == 64 ? (a->qkv_dtype ? pick_fwd<__bf16, 64>(njt) : pick_fwd<float, 64>(njt))
                                    : (a->qkv_dtype ? pick_fwd<__bf16, 32>(njt) : pick_fwd<float, 32>(njt));
    const size_t lds = fwd16_lds(njt, nw, a->dk);
    if (lds > 64 * 1024) ortk::lds_attr(reinterpret_cast<const void*>(fn), 160 * 1024);
    hipLaunchKernelGGL(fn, dim3((unsigned)(a->nkv * a->H)), dim3(64 * nw), lds, s, *a);
    ORTK_CHECK_LAUNCH();
    return 0;
}

int attn16_bwd(const ortk_attn_args* a, hipStream_t s) {
    const int njt = (a->Lk + 15) / 16, nw = (a->Lq + 15) / 16, Lqp = (int)ortk_align(a->Lq, 32);
    const bwd16_fn fn = a->dk == 64 ? (a->qkv_dtype ? pick_bwd<__bf16, 64>(njt) : pick_bwd<float, 64>(njt))
                                    : (a->qkv_dtype ? pick_bwd<__bf16, 32>(njt) : pick_bwd<float, 32>(njt));
    const size_t lds = bwd16_lds(njt, Lqp, a->dk);
    if (lds > 160 * 1024) return ORTK_EINVAL;
    if (lds > 64 * 1024) ortk::lds_attr(reinterpret_cast<const void*>(fn), 160 * 1024);
    hipLaunchKernelGGL(fn, dim3((unsigned)(a->nkv * a->H)), dim3(64 * nw), lds, s, *a, Lqp);
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace ortk
