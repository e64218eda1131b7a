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
#include "ortk_common.h"

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
                if (a.drop_p > 0.f) p = ortk_keep(a.drop_seed, (uint64_t)(pbase + j), a.drop_p) ? p * inv_keep : 0.f;
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

}  // namespace

extern "C" int ortk_attention_fwd(const ortk_attn_args* a, ortk_stream stream) {
    if (int e = check(a)) return e;
    if (!a->o) return ORTK_EINVAL;
    if (a->nkv == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fwd_lds_bytes(MAXK));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * 4 * (2 * 64 * KP + 128)));
        attr_set = true;
    }
    if (a->Lk <= 64) {
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
    if (!a->p || !a->d_o || !a->dq || !a->d_k || !a->dv || a->kv_index || a->kv_group_stride) return ORTK_EINVAL;
    if (a->nkv == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bwd_lds_bytes(MAXK));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_wave_kernel<64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * (2 * 64 * KP + 192)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_wave_kernel<48>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * (2 * 48 * KP + 192)));
        attr_set = true;
    }
    const int pairs = a->nkv * a->H;
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
